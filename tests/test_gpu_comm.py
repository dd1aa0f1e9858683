"""The library's RCCL communicator (csrc/comm.hip) on the hardware available to the tests: ONE GPU, so world = 1 -- every entry point
runs through librccl (unique id, ncclCommInitRank, ncclAllGather, grouped send / recv with only the self slice), and the C++ host
prover sharded over it emits the unsharded proof.  World > 1 needs one GPU per rank (RCCL refuses two ranks on a device); the same
sharding logic is covered with gloo on CPU (tests/test_dist_cpu.py) and with two gloo ranks sharing the GPU (tests/test_plonk.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_comm_world1_and_sharded_native_prover(hip, golden_srs):
    from ezkl_amd import backend as B, native as NV, plonk as P
    import test_plonk as TP
    assert B.comm_info() == (0, 0)
    B.comm_init(B.comm_unique_id(), 1, 0)
    try:
        assert B.comm_info() == (1, 0)
        pts = np.ascontiguousarray(golden_srs["g"][:5])
        assert (B.comm_fold_points(pts) == pts).all()                      # one rank: the fold is the identity
        buf = B.DeviceBuffer.from_numpy(np.arange(1024, dtype=np.uint64))
        B.comm_allgather_dev(buf.ptr, buf.nbytes)
        assert (buf.to_numpy() == np.arange(1024, dtype=np.uint64)).all()
        dst = B.DeviceBuffer(buf.nbytes)
        B.comm_alltoall_dev(buf.ptr, [256], [4096], dst.ptr, [512], [4096])
        got = dst.to_numpy()
        assert (got[64:64 + 512] == np.arange(32, 32 + 512, dtype=np.uint64)).all()
        # the multi-segment all-to-all and the host all_gather (world = 1: the self segments, matched by order)
        a, b_ = B.DeviceBuffer.from_numpy(np.arange(100, 164, dtype=np.uint64)), B.DeviceBuffer.from_numpy(np.arange(500, 532, dtype=np.uint64))
        ra, rb = B.DeviceBuffer(64 * 8), B.DeviceBuffer(32 * 8)
        B.comm_alltoallv_dev([(0, a.ptr, 64 * 8), (0, b_.ptr, 32 * 8)], [(0, ra.ptr, 64 * 8), (0, rb.ptr, 32 * 8)])
        assert (ra.to_numpy() == np.arange(100, 164, dtype=np.uint64)).all() and (rb.to_numpy() == np.arange(500, 532, dtype=np.uint64)).all()
        with pytest.raises(Exception):                                     # mismatched sizes are refused, not truncated
            B.comm_alltoallv_dev([(0, a.ptr, 64 * 8)], [(0, ra.ptr, 32 * 8)])
        h = np.arange(12, dtype=np.uint64).reshape(1, 12)
        assert (B.comm_allgather_host(h.copy()) == h).all()
        # per peer the segments are ONE byte stream: the two sides may cut it differently
        c3 = [B.DeviceBuffer(16 * 8), B.DeviceBuffer(48 * 8), B.DeviceBuffer(32 * 8)]
        B.comm_alltoallv_dev([(0, a.ptr, 64 * 8), (0, b_.ptr, 32 * 8)], [(0, c3[0].ptr, 16 * 8), (0, c3[1].ptr, 48 * 8), (0, c3[2].ptr, 32 * 8)])
        assert (np.concatenate([c.to_numpy() for c in c3]) == np.concatenate([np.arange(100, 164, dtype=np.uint64), np.arange(500, 532, dtype=np.uint64)])).all()
        # the shape of the prover's exchange -- hundreds of segments of very different sizes per peer -- through the PACKED wire format and
        # RCCL (EZKL_COMM_SELF_VIA_RCCL: the bytes to self take the slab + ncclSend / ncclRecv path instead of the copy kernel):
        # ONE send and ONE receive per peer and round, whatever the number of segments
        rng = np.random.default_rng(5)
        sizes = [int(x) * 32 for x in rng.integers(1, 1 << 15, 300)] + [1 << 24, 32, 1 << 22]
        src = [B.DeviceBuffer.from_numpy(rng.integers(0, 1 << 63, sz // 8, dtype=np.uint64)) for sz in sizes]
        total = sum(sizes)
        for slab_mb, unpacked in ((None, False), (16, False), (None, True)):
            dst = [B.DeviceBuffer(sz) for sz in sizes]
            os.environ["EZKL_COMM_SELF_VIA_RCCL"] = "1"
            if slab_mb: os.environ["EZKL_COMM_SLAB_MB"] = str(slab_mb)
            if unpacked: os.environ["EZKL_COMM_UNPACKED"] = "1"
            B.comm_stats(reset=True)
            try:
                B.comm_alltoallv_dev([(0, s_.ptr, sz) for s_, sz in zip(src, sizes)], [(0, d_.ptr, sz) for d_, sz in zip(dst, sizes)])
            finally:
                for v in ("EZKL_COMM_SELF_VIA_RCCL", "EZKL_COMM_SLAB_MB", "EZKL_COMM_UNPACKED"):
                    os.environ.pop(v, None)
            assert all((s_.to_numpy() == d_.to_numpy()).all() for s_, d_ in zip(src, dst)), (slab_mb, unpacked)
            st = B.comm_stats()
            rounds = -(-total // ((slab_mb or 128) << 20))
            if unpacked:
                assert st["nccl_sends"] == len(sizes) and st["nccl_recvs"] == len(sizes)           # round 3's wire format: one per segment
            else:
                assert st["nccl_sends"] == rounds and st["nccl_recvs"] == rounds and st["rounds"] == rounds, (st, rounds)
            # different cuts on the two sides, through the slabs: the receive side takes the same bytes as 7 equal pieces + the rest
        piece = (total // 7) & ~31
        cuts = [piece] * 7 + [total - 7 * piece]
        dst2 = [B.DeviceBuffer(sz) for sz in cuts]
        os.environ["EZKL_COMM_SELF_VIA_RCCL"] = "1"; os.environ["EZKL_COMM_SLAB_MB"] = "8"
        try:
            B.comm_alltoallv_dev([(0, s_.ptr, sz) for s_, sz in zip(src, sizes)], [(0, d_.ptr, sz) for d_, sz in zip(dst2, cuts)])
        finally:
            del os.environ["EZKL_COMM_SELF_VIA_RCCL"]; del os.environ["EZKL_COMM_SLAB_MB"]
        assert (np.concatenate([d_.to_numpy() for d_ in dst2]) == np.concatenate([s_.to_numpy() for s_ in src])).all()
        # segments of ANY address and length take the same packed wire format (an unaligned segment used to switch THIS rank to the
        # per-segment format while its aligned peers kept the slabs -- a hang): odd lengths at odd offsets through the slabs and RCCL
        raw = B.DeviceBuffer.from_numpy(rng.integers(0, 255, 1 << 16, dtype=np.uint8))
        out_ = B.DeviceBuffer.from_numpy(np.zeros(1 << 16, np.uint8))
        segs = [(3, 1001), (1004 + 13, 7), (5000, 4096 + 5), (20001, 33)]                 # (offset, bytes): nothing 16-byte aligned
        os.environ["EZKL_COMM_SELF_VIA_RCCL"] = "1"
        B.comm_stats(reset=True)
        try:
            B.comm_alltoallv_dev([(0, raw.ptr + o, n_) for o, n_ in segs], [(0, out_.ptr + o + 1, n_) for o, n_ in segs])
        finally:
            del os.environ["EZKL_COMM_SELF_VIA_RCCL"]
        st = B.comm_stats()
        assert st["nccl_sends"] == 1 and st["nccl_recvs"] == 1 and st["rounds"] == 1, st          # the slab format, not one send per segment
        a_, o_ = raw.to_numpy(dtype=np.uint8), out_.to_numpy(dtype=np.uint8)
        for o, n_ in segs:
            assert (o_[o + 1:o + 1 + n_] == a_[o:o + n_]).all(), (o, n_)
        # the C++ host prover over the library communicator: same bytes as the unsharded prover
        cs = TP.lookup_circuit(6)
        adv, fixed, copies = TP.lookup_witness(cs, 4)
        bg, bgl = B.Bases(golden_srs["g"]), B.Bases(golden_srs["g_lagrange"])
        plain = NV.NativeProvingKey(NV.NativeCircuit(cs), bg, fixed, copies)
        want = NV.create_proof(plain, bg, bgl, adv, seed=9)
        nc = NV.NativeCircuit(cs)
        assert nc.set_shard_comm() == (0, cs.n)
        nc.set_shard_full_bases(True)                                      # world 1 stays the plain prover; the owner mode needs world > 1
        pk = NV.NativeProvingKey(nc, bg, fixed, copies)
        assert NV.create_proof(pk, bg, bgl, adv, seed=9) == want
        # seed 0 on a sharded prover: rank 0's 256-bit OS-entropy key is broadcast over the communicator (not a 64-bit seed)
        from oracle import verifier as V
        p0, p1 = NV.create_proof(pk, bg, bgl, adv, seed=0), NV.create_proof(pk, bg, bgl, adv, seed=0)
        assert p0 != p1 and len(p0) == len(want)
        _, vk = P.keygen(cs, P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], 6), fixed, copies)
        g1, g2, s_g2 = TP.setup(golden_srs)
        assert V.verify(vk, g1, g2, s_g2, p0)
    finally:
        B.comm_destroy()
    assert B.comm_info() == (0, 0)


def test_comm_selftest_and_watchdog_world1(hip):
    """(a) the communicator tries itself out -- ezkl_hip_comm_init runs ezkl_hip_comm_selftest (all_gather of rank ids, ONE
    packed all-to-all of odd-sized segments, one fold of partial points) and it can be called again, also with the self bytes forced
    through the slabs and ncclSend / ncclRecv; (b) every collective waits under a watchdog: with a deadline no transfer can meet the
    call returns EZKL_ERR_TIMEOUT (-7) instead of hanging, later calls fail at once, and the process can destroy the communicator and
    build a new one.  In a subprocess: the test poisons its communicator on purpose."""
    import subprocess, sys
    from conftest import ROOT
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
B.comm_init(B.comm_unique_id(), 1, 0)            # runs the self-test
B.comm_selftest()
os.environ["EZKL_COMM_SELF_VIA_RCCL"] = "1"       # ... and once with the self bytes through the packed wire format and RCCL
B.comm_selftest()
os.environ.pop("EZKL_COMM_SELF_VIA_RCCL")
sends_before = B.comm_stats()["nccl_sends"] if isinstance(B.comm_stats(), dict) and "nccl_sends" in B.comm_stats() else None
pts = np.zeros((200000, 8), np.uint64)            # 12.8 MB up, gathered, 12.8 MB down: milliseconds
os.environ["EZKL_COMM_TIMEOUT_S"] = "0.000001"
try:
    B.comm_fold_points(pts)
    print("NO-TIMEOUT")
except ezkl_amd.EzklHipError as e:
    print("TIMEOUT-CODE", e.code if hasattr(e, "code") else e.args[0])
try:
    B.comm_allgather_host(np.zeros((1, 4), np.uint64))
    print("STILL-USABLE")
except ezkl_amd.EzklHipError as e:
    print("BROKEN-CODE", e.code if hasattr(e, "code") else e.args[0])
os.environ.pop("EZKL_COMM_TIMEOUT_S")
B.comm_destroy()
B.comm_init(B.comm_unique_id(), 1, 0)
assert (B.comm_fold_points(pts[:3]) == 0).all()
B.comm_destroy()
print("REBUILT")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "TIMEOUT-CODE -7" in out and "BROKEN-CODE -7" in out and "REBUILT" in out and "NO-TIMEOUT" not in out, out + r.stderr[-1500:]
    assert "did not complete within" in r.stderr and "fold_points" in r.stderr
