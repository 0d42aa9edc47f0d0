"""Child process of tests/test_fake_rccl.py: drives the C ABI's RCCL entry points (lc_comm_*, lc_halo_exchange,
lc_comm_flag_allreduce) as ranks 0..n-1 of n INSIDE ONE PROCESS ON ONE GPU, against the recording loopback stand-in
tests/c/fake_rccl.c, which this process loads before liblcs_hip.so (halo.hip's load_rccl prefers a copy of
librccl.so.1 the process already holds).  ctypes + numpy only -- importing torch would bring the real RCCL in first.

    python tests/fake_rccl_driver.py <path to the fake librccl.so.1>

Prints "fake-rccl driver: all checks passed" and exits 0, or raises."""
import ctypes as C
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "torch" not in sys.modules

fake = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)             # first: SONAME librccl.so.1 is now in the process
fake.fake_rccl_log_line.restype = C.c_char_p
from lagrangiancoherence_amd import _capi, flows, sharded  # noqa: E402  (none of these import torch)
lib = _capi.load(import_torch=False)
assert "torch" not in sys.modules
hip = fake                                                # dlsym through the stand-in reaches its dependency libamdhip64 (the one copy in the process)
LC_F32, LC_F64, LC_ERCCL = _capi.LC_F32, _capi.LC_F64, _capi.LC_ERCCL
NCCL_DT = {LC_F32: 7, LC_F64: 8}


def ck(st):
    _capi.check(st, lib)


def log(reset=True):
    out = [fake.fake_rccl_log_line(i).decode() for i in range(fake.fake_rccl_nlog())]
    if reset:
        fake.fake_rccl_reset()
    return out


def field(line, key):
    return int(re.search(rf"\b{key}=(\d+)", line).group(1))


class Dev:
    """Device buffer through lc_malloc / lc_memcpy_*."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes, self.p = ctx, nbytes, C.c_void_p()
        ck(lib.lc_malloc(ctx, nbytes, C.byref(self.p)))

    def up(self, a):
        a = np.ascontiguousarray(a)
        ck(lib.lc_memcpy_h2d(self.ctx, self.p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return self

    def down(self, shape, dtype):
        a = np.empty(shape, dtype)
        ck(lib.lc_memcpy_d2h(self.ctx, a.ctypes.data_as(C.c_void_p), self.p, a.nbytes))
        return a

    def at(self, byte_off):
        return C.c_void_p(self.p.value + byte_off)


def make_ranks(world):
    ctxs, streams = [], []
    for r in range(world):
        ctx = C.c_void_p()
        ck(lib.lc_ctx_create(0, C.byref(ctx)))
        st = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(st)) == 0
        ck(lib.lc_ctx_set_stream(ctx, st))                # so the test knows which stream every rank's calls must use
        ctxs.append(ctx)
        streams.append(st.value)
    uid = C.create_string_buffer(128)
    ck(lib.lc_comm_unique_id(uid, 128))
    comms = []
    for r in range(world):
        c = C.c_void_p()
        ck(lib.lc_comm_create(ctxs[r], world, r, uid, 128, C.byref(c)))
        n, rr = C.c_int(), C.c_int()
        ck(lib.lc_comm_count(c, C.byref(n), C.byref(rr)))
        assert (n.value, rr.value) == (world, r)
        comms.append(c)
    lines = log()
    assert lines[0].startswith("GetUniqueId") and [ln for ln in lines if ln.startswith("CommInitRank")] == \
        [f"CommInitRank id={field(lines[0], 'id')} nranks={world} rank={r}" for r in range(world)], lines
    return ctxs, streams, comms


def halo_case(world, dtype):
    """Row-sharded advection of one small grid as `world` ranks, halo rows through lc_halo_exchange: every call's
    (peer, byte offset, count, dtype, stream) inside one GroupStart/End, and the exchanged buffers equal, bit for bit,
    the same rows of the unsharded result."""
    npdt = np.float32 if dtype == LC_F32 else np.float64
    es = np.dtype(npdt).itemsize
    u, v, lat, lon = flows.era5_like(nt=5, ny=72, nx=144)
    u, v, lat, lon = (a.astype(npdt) for a in (u, v, lat, lon))
    slat, slon = flows.seed_grid(101, 160, lat, lon)
    slat, slon = slat.astype(npdt), slon.astype(npdt)
    nyg, nx, nt, (ny_f, nx_f) = slat.size, slon.size, u.shape[0], u.shape[1:]
    ctxs, streams, comms = make_ranks(world)

    def advect(ctx, lo, hi, x_dst, y_dst):
        ud, vd = Dev(ctx, u.nbytes).up(u), Dev(ctx, v.nbytes).up(v)
        n = lib.lc_packed_elems(nt, ny_f, nx_f)
        lin, ext = Dev(ctx, n * es), Dev(ctx, lib.lc_packed_elems(nt - 1, ny_f, nx_f) * es)
        ck(lib.lc_field_pack(ctx, ud.p, vd.p, dtype, nt, ny_f, nx_f, 1, lin.p, ext.p))
        sl, so = Dev(ctx, (hi - lo) * es).up(slat[lo:hi]), Dev(ctx, nx * es).up(slon)
        ck(lib.lc_advect(ctx, lin.p, None, ext.p, dtype, nt, ny_f, nx_f, float(lat[0]), float(lat[-1]), float(lon[0]),
                         float(lon[-1]), sl.p, hi - lo, so.p, nx, lo, nyg, -900.0, 4, 1, _capi.LC_X_CYCLIC, 0, nt - 1,
                         x_dst, y_dst, None, None))
        ck(lib.lc_sync(ctx))
        for d in (ud, vd, lin, ext, sl, so):
            lib.lc_free(ctx, d.p)

    xf, yf = Dev(ctxs[0], nyg * nx * es), Dev(ctxs[0], nyg * nx * es)
    advect(ctxs[0], 0, nyg, xf.p, yf.p)
    x_full, y_full = xf.down((nyg, nx), npdt), yf.down((nyg, nx), npdt)
    bufs, geo = [], []
    for r in range(world):
        lo, hi = sharded.row_partition(nyg, world, r)
        n_lo, n_hi = sharded.halo_rows(nyg, lo, hi)
        rows = n_lo + (hi - lo) + n_hi
        nan = np.full((rows, nx), np.nan, npdt)
        xe, ye = Dev(ctxs[r], nan.nbytes).up(nan), Dev(ctxs[r], nan.nbytes).up(nan)
        advect(ctxs[r], lo, hi, xe.at(n_lo * nx * es), ye.at(n_lo * nx * es))
        bufs.append((xe, ye))
        geo.append((lo, hi, n_lo, n_hi, rows))
    fake.fake_rccl_reset()
    for r in range(world):                                        # rank after rank: the stand-in matches sends and receives
        lo, hi, n_lo, n_hi, rows = geo[r]
        xe, ye = bufs[r]
        ck(lib.lc_halo_exchange(ctxs[r], comms[r], xe.p, ye.p, dtype, rows, nx, n_lo, n_hi))
        lines = log(reset=False)[-(2 + 4 * ((r > 0) + (r < world - 1))):] if world > 1 else []
        if world > 1:
            n, row = hi - lo, nx * es
            want = []
            for b in (xe, ye):
                if r > 0:
                    want += [("Send", r - 1, b.p.value + n_lo * row), ("Recv", r - 1, b.p.value)]
                if r < world - 1:
                    want += [("Send", r + 1, b.p.value + (n_lo + n - 2) * row), ("Recv", r + 1, b.p.value + (n_lo + n) * row)]
            assert lines[0] == "GroupStart" and lines[-1] == "GroupEnd", lines
            got = [(ln.split()[0], field(ln, "peer"), field(ln, "ptr")) for ln in lines[1:-1]]
            assert got == want, (r, got, want)
            for ln in lines[1:-1]:
                assert field(ln, "rank") == r and field(ln, "count") == 2 * nx and field(ln, "dtype") == NCCL_DT[dtype] \
                    and field(ln, "stream") == streams[r] and field(ln, "in_group") == 1, ln
    assert fake.fake_rccl_pending() == 0 and fake.fake_rccl_group_depth() == 0       # every send met its receive
    for r in range(world):
        ck(lib.lc_sync(ctxs[r]))
    for r in range(world):
        lo, hi, n_lo, n_hi, rows = geo[r]
        xe, ye = bufs[r][0].down((rows, nx), npdt), bufs[r][1].down((rows, nx), npdt)
        assert np.array_equal(xe, x_full[lo - n_lo:hi + n_hi]) and np.array_equal(ye, y_full[lo - n_lo:hi + n_hi]), \
            f"rank {r} of {world}: exchanged rows differ from the unsharded result"
        assert not np.isnan(xe).any()
    # ---- wrong halo sizes are refused before any RCCL call
    fake.fake_rccl_reset()
    if world > 1:
        lo, hi, n_lo, n_hi, rows = geo[0]
        assert lib.lc_halo_exchange(ctxs[0], comms[0], bufs[0][0].p, bufs[0][1].p, dtype, rows, nx, 2, n_hi) == _capi.LC_EINVAL
        assert log() == []
    # ---- an injected failure inside the group: LC_ERCCL, the group is still closed, the message names RCCL
    if world > 1:
        for fail_at in (1, 2, 3):
            r = 1 if world > 2 else 0
            lo, hi, n_lo, n_hi, rows = geo[r]
            fake.fake_rccl_reset()
            fake.fake_rccl_fail_at(fail_at)
            st = lib.lc_halo_exchange(ctxs[r], comms[r], bufs[r][0].p, bufs[r][1].p, dtype, rows, nx, n_lo, n_hi)
            lines = log()
            assert st == LC_ERCCL and b"RCCL send/recv failed" in lib.lc_last_error(), (st, lib.lc_last_error())
            assert lines[0] == "GroupStart" and lines[-1] == "GroupEnd" and "INJECTED FAILURE" in lines, lines
            assert sum(ln.startswith(("Send", "Recv")) for ln in lines) == fail_at       # nothing posted after the failure
            assert fake.fake_rccl_group_depth() == 0
    # ---- lc_comm_flag_allreduce: uint32 max, in place, on the stream of the communicator's context
    fake.fake_rccl_reset()
    count = nx
    flags = []
    for r in range(world):
        f = np.zeros(count, np.uint32)
        f[r::world + 1] = 1
        flags.append((f, Dev(ctxs[r], f.nbytes).up(f)))
    for r in range(world):
        ck(lib.lc_comm_flag_allreduce(comms[r], flags[r][1].p, count))
    lines = log()
    if world == 1:
        assert lines == []                                             # a 1-rank communicator needs no collective
    else:
        assert len(lines) == world
        for r, ln in enumerate(lines):
            assert ln.startswith("AllReduce") and field(ln, "rank") == r and field(ln, "count") == count and field(ln, "dtype") == 3 \
                and field(ln, "op") == 2 and field(ln, "stream") == streams[r] and field(ln, "send") == field(ln, "recv") == flags[r][1].p.value, ln
        want = np.maximum.reduce([f for f, _ in flags])
        for r in range(world):
            assert np.array_equal(flags[r][1].down((count,), np.uint32), want)
        assert fake.fake_rccl_pending() == 0
    assert lib.lc_comm_flag_allreduce(comms[0], None, 0) == 0 and lib.lc_comm_flag_allreduce(None, None, 4) == _capi.LC_EINVAL
    for r in range(world):
        lib.lc_comm_destroy(comms[r])
        lib.lc_ctx_destroy(ctxs[r])
    assert sum(ln.startswith("CommDestroy") for ln in log()) == world


if __name__ == "__main__":
    for world in (1, 2, 3):
        for dtype in (LC_F32, LC_F64):
            halo_case(world, dtype)
            print(f"fake-rccl driver: world {world} dtype {'f32' if dtype == LC_F32 else 'f64'} ok", flush=True)
    print("fake-rccl driver: all checks passed")
