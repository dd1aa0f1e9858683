#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp; python "$R/tools/h2d_probe2.py" 2>&1 | grep "2^"
