"""CPU tests of the host reduction ring (phip_comm_allreduce_host): the shared-memory meeting point of the ranks' partial dot products
and norms on one node.  Threads of one process ("threads" backend on planning-only contexts) and processes ("shm" backend)."""
import ctypes as C
import os
import struct
import subprocess
import sys
import threading

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("n", [2, 3, 8])
def test_ring_threads(n):
    from petsc_amd import capi
    L = capi.load()
    key = struct.pack("<Q", 0xabc000 + n) + bytes(120)
    rounds = 3000
    table = np.random.default_rng(4).standard_normal((rounds, n, 3))
    err = [None] * n

    def work(r):
        try:
            ctx = C.c_void_p()
            assert L.phip_init(-1, C.byref(ctx)) == 0
            comm = C.c_void_p()
            assert L.phip_comm_create(ctx, b"threads", key, n, r, C.byref(comm)) == 0
            for k in range(rounds):
                v = table[k, r].copy()
                assert L.phip_comm_allreduce_host(comm, v.ctypes.data_as(C.POINTER(C.c_double)), 3, 0) == 0
                ref = table[k, 0].copy()
                for p in range(1, n):
                    ref += table[k, p]
                assert np.array_equal(v, ref)
            assert L.phip_comm_destroy(comm) == 0
            assert L.phip_finalize(ctx) == 0
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    for e in err:
        if e is not None:
            raise e


@pytest.mark.parametrize("n", [2, 4])
def test_ring_processes_shm(n):
    key = os.urandom(128).hex()
    env = dict(os.environ, PHIP_HOST_RING_TIMEOUT="60")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_ring_worker.py"), key, str(n), str(r), "2000"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(n)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "ring ok rank %d" % r in o, o
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("phip_")]  # the segment's name is gone after the first round


def test_ring_ignores_stale_segment():
    """A run that died during set-up leaves /dev/shm/phip_<hash> behind, ready-marked and with old sequence words.  A later run under
    the same key must not take it: ranks other than 0 accept a segment only while its creator is alive."""
    import struct
    n = 2
    key = os.urandom(128)
    h = 1469598103934665603
    for b in key:
        h = ((h ^ b) * 1099511628211) & (2**64 - 1)
    h = ((h ^ n) * 1099511628211) & (2**64 - 1)
    name = "/dev/shm/phip_%016x" % h
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    slot = 64 + 8 * 128
    seg = bytearray(64 + slot * 4 * n)
    struct.pack_into("<QQ", seg, 0, 0x70686970686f7374, dead.pid)
    for k in range(4 * n):
        struct.pack_into("<Q", seg, 64 + slot * k, 1)       # "round 1 already written" by everyone, with garbage payloads
        seg[64 + slot * k + 64:64 + slot * k + 72] = b"\xff" * 8
    with open(name, "wb") as f:
        f.write(seg)
    try:
        env = dict(os.environ, PHIP_HOST_RING_TIMEOUT="30")
        # rank 1 first, so that it meets the leftover before rank 0 replaces it
        p1 = subprocess.Popen([sys.executable, os.path.join(HERE, "_ring_worker.py"), key.hex(), str(n), "1", "50"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        import time
        time.sleep(1.0)
        p0 = subprocess.Popen([sys.executable, os.path.join(HERE, "_ring_worker.py"), key.hex(), str(n), "0", "50"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        outs = []
        for p in (p0, p1):
            try:
                o, _ = p.communicate(timeout=90)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
            outs.append(o)
        for r, (p, o) in enumerate(zip((p0, p1), outs)):
            assert p.returncode == 0 and "ring ok rank %d" % r in o, o
    finally:
        if os.path.exists(name):
            os.unlink(name)


def test_ring_missing_peer_times_out():
    """A rank whose peers never arrive gets an error, not a hang."""
    from petsc_amd import capi
    env = dict(os.environ, PHIP_HOST_RING_TIMEOUT="1")
    code = ("import ctypes as C,sys,os\nsys.path.insert(0,%r)\nfrom petsc_amd import capi\nL=capi.load()\nctx=C.c_void_p();L.phip_init(-1,C.byref(ctx))\n"
            "comm=C.c_void_p()\nrc=L.phip_comm_create(ctx,b'shm',os.urandom(128),2,0,C.byref(comm))\nprint('rc',rc,capi.last_error())\n" % os.path.dirname(HERE))
    o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert "rc 76" in o.stdout and "did not arrive" in o.stdout, o.stdout + o.stderr
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("phip_")]  # rank 0 removed the name on its way out
