#!/bin/bash
free -g | head -2; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null; df -h /tmp | tail -1
