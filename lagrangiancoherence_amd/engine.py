"""Array-level host API over the C ABI (include/lcs_hip.h).

`Engine` keeps wind fields, seeds and results resident in HBM as torch tensors
(torch is plumbing here: device memory, streams, torch.distributed) and calls
the HIP kernels through ctypes on torch's current stream.  `lcs_host` is the
torch-free route: host numpy arrays through the one-call entry point
`lc_lcs_host`.

Everything is arrays `(time, latitude, longitude)`, coordinates ascending; the
xarray-facing drop-in surface (`LagrangianCoherence.LCS.*`) sits on top of this
module in `dropin.py`.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _capi

__all__ = ["Engine", "PackedField", "lcs_host", "lcs_global_host", "common_dtype", "x_boundary_mode"]

_NP2LC = {np.dtype(np.float32): _capi.LC_F32, np.dtype(np.float64): _capi.LC_F64}
_LAYOUTS = {"reference": _capi.LC_LAYOUT_REFERENCE, "physical": _capi.LC_LAYOUT_PHYSICAL}


def x_boundary_mode(cyclic_xboundary, noncyclic_clamp=None, whole_grid=True) -> int:
    """lc_advect's ``cyclic_x`` (enum lc_xboundary).  ``cyclic_xboundary=False`` means the reference's own
    outer-product clamp (``noncyclic_clamp='reference_outer'``, LCS/trajectory.py:96-97, Q9).  The rule couples all
    seed rows through the offending columns, so a call on a row block (``whole_grid=False``) is only possible with a
    flag all-reduce over the ranks (:meth:`Engine.set_flag_allreduce`, which ``sharded_lcs`` installs) -- without one
    lc_advect refuses it; nothing is silently replaced by the per-point clamp (``noncyclic_clamp='pointwise'`` asks
    for that one explicitly; it differs from the reference only if a parcel leaves the longitude range)."""
    if cyclic_xboundary:
        return _capi.LC_X_CYCLIC
    if noncyclic_clamp is None:
        noncyclic_clamp = "reference_outer"
    if noncyclic_clamp not in ("pointwise", "reference_outer"):
        raise ValueError(f"noncyclic_clamp {noncyclic_clamp!r}: 'pointwise' or 'reference_outer'")
    return _capi.LC_X_CLAMP_REFERENCE_OUTER if noncyclic_clamp == "reference_outer" else _capi.LC_X_CLAMP_POINT


def common_dtype(*arrays) -> np.dtype:
    """float32 only if every input is float32, else float64.

    The reference lets numpy promote (fp32 wind on fp64 coordinates gives fp64
    positions and fp32 velocities); the kernels have one arithmetic
    type per call, so a mixed call is computed in float64.
    """
    dts = {np.dtype(str(getattr(a, "dtype", "float64")).replace("torch.", "")) for a in arrays if a is not None}
    return np.dtype(np.float32) if dts == {np.dtype(np.float32)} else np.dtype(np.float64)


@dataclass
class PackedField:
    """Gather-ready image(s) of a wind time series, resident on the device."""
    lin: "torch.Tensor | None"     # order-1 image; None when the raw planes ``u``, ``v`` below serve as the order-1 source
    cub: "torch.Tensor | None"     # B-spline coefficient image of order ``order`` (2..5), None for order 1
    ext: "torch.Tensor | None"     # 2*img[t]-img[t+1] of the image matching interp_order (fused SETTLS sample)
    nt: int
    ny_f: int
    nx_f: int
    lat_min: float
    lat_max: float
    lon_min: float
    lon_max: float
    dtype: np.dtype
    wind_f32: bool = False          # float32 wind on float64 coordinates: numpy's promotion rules in lc_advect
    order: int = 1                  # interpolation order the field was prepared for (order 1 is always available)
    fuse_raw: bool = False          # float64 at order 1: fused levels with NO packed image (2 F[t] - F[t+1] formed from u, v in the kernels)
    u: "torch.Tensor | None" = None  # the raw planes (nt, ny_f, nx_f) the images were packed from, kept as the ORDER-1 source
    v: "torch.Tensor | None" = None  # (lc_advect_ex: pole rows at any order, the Euler sample in float64) -- not copies: do not
    #                                  modify them in place while the field is in use (Engine._ensure_lin refuses if you did)
    planes_version: "tuple | None" = None  # (u._version, v._version) when the field was prepared
    lin32: "torch.Tensor | None" = None    # wind_f32 at order 1: the order-1 image of the float32 wind AS float32 (LC_F64_WIND_F32_LIN32)
    u32: "torch.Tensor | None" = None      # wind_f32: the float32 planes as given (the float64 copies u, v are made when a call needs them)
    v32: "torch.Tensor | None" = None
    planes32_version: "tuple | None" = None  # (u32._version, v32._version) when the field was prepared: borrowed like u, v


class Engine:
    """One context on one MI355X.  Not a CPU fallback: needs the HIP library and a GPU."""

    def __init__(self, device: int | None = None):
        import torch
        self.torch = torch
        self.lib = _capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("lagrangiancoherence_amd.Engine needs a GPU (torch.cuda.is_available() is False); "
                               "there is no CPU path")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        ctx = C.c_void_p()
        _capi.check(self.lib.lc_ctx_create(self.device_index, C.byref(ctx)), self.lib)
        self.lds_tiles_mode = -1      # what set_lds_tiles was last given (-1: the library's default, or LCS_LDS_TILES)
        self.verify_mode = 0          # lc_ctx_set_verify
        self._poison = bool(os.environ.get("LCS_DEBUG_POISON"))
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.lc_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lds_tiles(self, mode: int):
        """lc_advect float32 kernel choice: -1 default (LDS tiles; at order 1 two seeds per lane from 2^23 seeds per call,
        one below), 1 LDS tiles with two seeds per lane at order 1 whatever the size, 2 one seed per lane, 0 direct gathers."""
        _capi.check(self.lib.lc_ctx_set_lds_tiles(self.ctx, int(mode)), self.lib)
        self.lds_tiles_mode = int(mode)

    def set_sigma_march(self, on: int):
        """lc_sigma float32 kernel choice: -1 default (marching kernel with wavefront shuffles from 2^23 cells per call,
        LDS tiles below), 1 marching kernel whatever the size, 0 LDS tiles."""
        _capi.check(self.lib.lc_ctx_set_sigma_march(self.ctx, int(on)), self.lib)

    def set_level_chunk(self, levels: int):
        """Run every advect call as consecutive launches of at most ``levels`` time levels (0: one launch; -1: by size,
        the default -- 32 levels from 2^18 seeds per call).  Results are bit-identical; it shapes the launches only
        (``lc_ctx_set_level_chunk``)."""
        _capi.check(self.lib.lc_ctx_set_level_chunk(self.ctx, int(levels)), self.lib)

    @property
    def level_chunk(self) -> int:
        """The context's levels-per-launch setting in force (``lc_ctx_get_level_chunk``): what :meth:`set_level_chunk`
        was last given, or what ``LCS_LEVEL_CHUNK`` set when the context was created (-1: by size)."""
        v = C.c_int()
        _capi.check(self.lib.lc_ctx_get_level_chunk(self.ctx, C.byref(v)), self.lib)
        return v.value

    _FIDELITY = {"auto": _capi.LC_F64_AUTO, "exact": _capi.LC_F64_EXACT_ORDER, "fast": _capi.LC_F64_FAST}

    def set_f64_fidelity(self, mode: str):
        """float64 on the reference-shaped surfaces (the drop-in's ``LCS`` / ``parcel_propagation``, the one-call host
        routes): ``'exact'`` = numpy / scipy's operation order (~1e-13 degrees from the reference), ``'fast'`` = the
        fused-level form (rounding-level differences, <= 1e-9 degrees or the flow's own response to a 1e-12 degree seed
        shift, whichever is larger), ``'auto'`` (default) = exact up to 2^18 seeds per call, fast above
        (``lc_ctx_set_f64_fidelity``).  :meth:`prepare_field`'s own ``fuse_levels`` argument is explicit and unaffected."""
        if mode not in self._FIDELITY:
            raise ValueError(f"float64 fidelity {mode!r}: 'auto', 'exact' or 'fast'")
        _capi.check(self.lib.lc_ctx_set_f64_fidelity(self.ctx, self._FIDELITY[mode]), self.lib)

    def f64_fuse_levels(self, dtype, n_seeds: int) -> bool:
        """``fuse_levels`` for a reference-shaped call of ``n_seeds`` seeds in ``dtype`` under the context's fidelity
        setting (the rule of ``lc_lcs_host``: float32 always fuses)."""
        if np.dtype(dtype) != np.dtype(np.float64):
            return True
        m = C.c_int()
        _capi.check(self.lib.lc_ctx_get_f64_fidelity(self.ctx, C.byref(m)), self.lib)
        if m.value == _capi.LC_F64_EXACT_ORDER:
            return False
        return m.value == _capi.LC_F64_FAST or int(n_seeds) > _capi.LC_EXACT_ORDER_MAX_SEEDS

    def set_flag_allreduce(self, group=None, comm=None, enable=True):
        """Row-sharded grids with the reference's non-cyclic clamp (``LC_X_CLAMP_REFERENCE_OUTER``): install the
        MAX all-reduce of the offending-column flags over the ranks that share the seed grid
        (``lc_ctx_set_flag_allreduce``).  ``comm``: the C ABI's RCCL communicator (``lc_comm_flag_allreduce``);
        otherwise ``torch.distributed`` over ``group`` (nccl = RCCL reduces the device buffer in place on the current
        stream; gloo, used when rehearsing on one GPU, goes through the host).  ``enable=False`` removes it."""
        if not enable:
            _capi.check(self.lib.lc_ctx_set_flag_allreduce(self.ctx, None, None), self.lib)
            self._flag_cb = None
            return
        if comm is not None:
            fn = C.cast(self.lib.lc_comm_flag_allreduce, C.c_void_p)
            _capi.check(self.lib.lc_ctx_set_flag_allreduce(self.ctx, fn, comm), self.lib)
            self._flag_cb = None
            return
        torch = self.torch
        import torch.distributed as dist

        class _DeviceWords:     # a raw device pointer as a tensor, through the CUDA array interface
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}

        def reduce(_user, ptr, count):
            try:
                t = torch.as_tensor(_DeviceWords(ptr, count), device=self.device)     # the flags are 0 / 1: int32 max is the OR
                if dist.get_backend(group) == "gloo":
                    h = t.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                return 0
            except Exception:     # an exception must not unwind through the C frames of lc_advect
                import traceback
                traceback.print_exc()
                return 1
        self._flag_cb = _capi.FLAG_ALLREDUCE_FN(reduce)      # kept alive for as long as the context may call it
        _capi.check(self.lib.lc_ctx_set_flag_allreduce(self.ctx, C.cast(self._flag_cb, C.c_void_p), None), self.lib)

    def last_advect_kernel(self) -> str:
        """Name of the kernel the last :meth:`advect` call launched (as a profiler shows it)."""
        return self.lib.lc_ctx_last_advect_kernel(self.ctx).decode()

    def set_verify(self, mode: int = 1):
        """Wave-state audit of the one-seed order-1 LDS kernel (``lc_ctx_set_verify``): 1 on, 2 on with one injected
        corruption (test hook), 0 off.  Results are bit-identical; :meth:`read_verify` returns the counters."""
        _capi.check(self.lib.lc_ctx_set_verify(self.ctx, int(mode)), self.lib)
        self.verify_mode = int(mode)

    def read_verify(self, reset: bool = True) -> dict:
        """The audit's counters (synchronises): ``tile_changed`` wave-levels whose LDS tile no longer held what the wave
        staged, ``entries_changed`` 16-byte entries, ``slot_changes`` wave-levels at which the wave sat in another hardware
        slot than a level earlier (its context was switched out and back in), ``audited`` wave-levels checked, and the
        first event (workgroup, tile, wave, level, HW_ID before / after, lane mask)."""
        out = (C.c_uint * 16)()
        _capi.check(self.lib.lc_ctx_read_verify(self.ctx, out, int(bool(reset))), self.lib)
        v = [int(x) for x in out]
        d = {"tile_changed": v[0], "entries_changed": v[1], "slot_changes": v[2], "audited": v[3]}
        if v[12]:
            d["first_event"] = {"workgroup": v[4], "tile": v[5], "wave": v[6], "level": v[7], "hw_id_before": hex(v[8]),
                                "hw_id_after": hex(v[9]), "lane_mask": hex(v[10] | (v[11] << 32))}
        return d

    def last_advect_launches(self) -> int:
        """Kernel launches the last :meth:`advect` call made (level chunks)."""
        return int(self.lib.lc_ctx_last_advect_launches(self.ctx))

    TWO_SEED_MIN = 1 << 23   # seeds per call from which lc_advect's default is the two-seeds-per-lane kernel
    # float64 at order 1 with fused levels: build the fused-level image (True), or let the kernels form it from the raw
    # planes (False: no pack at all).  Measured on BASELINE configs[1] (profiles/r04): see DESIGN.md section 4.
    EXT_IMAGE_F64 = True
    # float64 at order 3 with fused levels: build the fused-level COEFFICIENT image ext = 2 cub[t] - cub[t+1] (True), or let
    # the kernels form it from cub node by node (False: the pack neither reads the coefficients back nor writes a second
    # image: 4.56 -> 3.28 ms on BASELINE configs[1]; but the advect kernel then stages two levels per iteration tile and
    # takes 8.22 ms instead of 6.31 -- the step loses 0.6 ms either way, serial and pipelined: profiles/r05/c2_o3_no_ext_image_ab.txt).
    EXT_IMAGE_F64_O3 = True

    class _Concurrent:
        """Context manager for ``n`` advect calls running side by side on different streams: what fills the machine is
        the seeds in flight, so the size rule of the default kernel choice is applied to ``n`` calls' worth."""

        def __init__(self, eng, seeds_per_call, n):
            self.eng, self.force = eng, eng.lds_tiles_mode == -1 and "LCS_LDS_TILES" not in os.environ \
                and seeds_per_call < Engine.TWO_SEED_MIN <= seeds_per_call * n

        def __enter__(self):
            if self.force:
                self.eng.set_lds_tiles(1)
            return self

        def __exit__(self, *exc):
            if self.force:
                self.eng.set_lds_tiles(-1)
            return False

    def concurrent_calls(self, seeds_per_call: int, n: int):
        return Engine._Concurrent(self, int(seeds_per_call), int(n))

    def last_sigma_kernel(self) -> str:
        """Name of the kernel the last :meth:`sigma` / :meth:`flowmap_gradient` call launched."""
        return self.lib.lc_ctx_last_sigma_kernel(self.ctx).decode()

    # ------------------------------------------------------------------ plumbing
    def last_pack_kernel(self) -> str:
        """The kernel the last ``lc_field_pack`` launched for its interleave / prefilter stage (``lc_ctx_last_pack_kernel``)."""
        return self.lib.lc_ctx_last_pack_kernel(self.ctx).decode()

    def _use_current_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        _capi.check(self.lib.lc_ctx_set_stream(self.ctx, C.c_void_p(s)), self.lib)

    STAGED_COPY_FROM = 8 << 20   # bytes from which host <-> device copies go through the context's pinned staging ring

    def to_device(self, a, dtype: np.dtype):
        torch = self.torch
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=getattr(torch, np.dtype(dtype).name))
            return t.contiguous()
        a = np.asarray(a)
        if a.ndim > 1 and not a.flags.c_contiguous and a.dtype == np.dtype(dtype) and a.nbytes >= self.STAGED_COPY_FROM:
            # a transposed VIEW of a contiguous array (the reference's example builds u, v with dims (latitude, longitude,
            # time) and the adapter asks for (time, latitude, longitude): LCS/LCS.py:101-104): making it contiguous on the host
            # is a strided copy at ~1 GB/s.  The buffer travels as it lies in memory, the permutation runs on the device.
            order = sorted(range(a.ndim), key=lambda i: -abs(a.strides[i]))
            base = a.transpose(order)
            if base.flags.c_contiguous:
                inv = [order.index(i) for i in range(a.ndim)]
                return self.to_device(base, dtype).permute(inv).contiguous()
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.nbytes >= self.STAGED_COPY_FROM:
            # a large pageable array (a reanalysis wind series): through the ring of pinned buffers, at the bus rate whatever
            # state its pages are in (lc_copy_to_device; a first-touch pageable copy runs at a quarter of it)
            t = torch.empty(a.shape, dtype=getattr(torch, np.dtype(dtype).name), device=self.device)
            self._use_current_stream()
            _capi.check(self.lib.lc_copy_to_device(self.ctx, C.c_void_p(t.data_ptr()), a.ctypes.data_as(C.c_void_p), a.nbytes), self.lib)
            return t
        return torch.from_numpy(a).to(self.device).contiguous()

    def to_host(self, t) -> np.ndarray:
        """A device tensor as a numpy array (results of the drop-in surface); large ones through the staging ring."""
        t = t.detach()
        if not t.is_cuda or t.numel() * t.element_size() < self.STAGED_COPY_FROM:
            return t.cpu().numpy()
        t = t.contiguous()
        out = np.empty(tuple(t.shape), dtype=np.dtype(str(t.dtype).replace("torch.", "")))
        self._use_current_stream()
        _capi.check(self.lib.lc_copy_to_host(self.ctx, out.ctypes.data_as(C.c_void_p), C.c_void_p(t.data_ptr()), out.nbytes), self.lib)
        return out

    def _empty(self, shape, dtype):
        t = self.torch.empty(shape, dtype=getattr(self.torch, np.dtype(dtype).name), device=self.device)
        if self._poison and t.is_floating_point():
            # LCS_DEBUG_POISON=1 (read once, at Engine creation): every buffer the engine allocates starts as NaN instead of
            # whatever the caching allocator hands back -- typically the previous call's identical results, which would make
            # an element that a kernel forgot to write look right.  The GPU suite runs green with it.
            t.fill_(float("nan"))
        return t

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    # ------------------------------------------------------------------ field
    def prepare_field(self, u, v, lat_f, lon_f, interp_order: int = 1, dtype=None, fuse_levels=None,
                      lin_image=None, ext_image=None) -> PackedField:
        """Upload (if needed) and pack a wind series.  u, v: (nt, ny_f, nx_f).

        ``lin_image``: build the order-1 image too.  Default (None): only where a kernel reads it -- float32 at
        ``interp_order=1``.  Everywhere else the raw planes themselves are the order-1 source (``lc_advect_ex``: the
        pole seed rows at any order, LCS/tools.py:31-39, and in float64 the Euler sample): the field keeps a reference to
        the device copies of ``u`` and ``v`` instead of a second, interleaved copy, the pack writes a third (order 1) to a
        fifth (order 3) fewer bytes, and results are bit-identical either way.

        ``ext_image`` (float64 at ``interp_order=1`` with ``fuse_levels``): False = do not build the fused-level image
        either -- the kernels form ``2 F[t] - F[t+1]`` from the raw planes node by node (``lc_advect_args.fuse_levels_raw``;
        the same expression, bit-identical results): such a field needs NO pack at all.  Default: see ``EXT_IMAGE_F64``.
        At ``interp_order=3`` (float64): False = the kernels form the fused-level coefficients ``2 cub[t] - cub[t+1]`` from
        the coefficient image node by node (bit-identical too): the pack is the two prefilter sweeps and the pads, nothing
        more.  Default: see ``EXT_IMAGE_F64_O3``.

        ``fuse_levels``: also build ext[t] = 2 F[t] - F[t+1] so each SETTLS iteration takes one
        gather instead of two (interpolation is linear in the field => the same value up to rounding).
        Default: on, for float32 and -- since round 3 -- float64 (orders 1 and 3; float64 positions move by
        <= 1e-10 degrees against the reference's operation order on config 2, inside the 1e-9 degrees the float64
        parity tests state).  ``fuse_levels=False`` keeps numpy / scipy's exact operation order in float64
        (two samples per iteration, true divisions, scipy's tap sum): results equal to the CPU oracle's to
        ~1e-13 degrees, at 1.4x the time."""
        if interp_order not in (1, 2, 3, 4, 5):
            raise ValueError(f"interp_order {interp_order} unsupported (scipy's spline orders 1..5; "
                             "0 fails in the reference too, LCS/tools.py:24-30)")
        lat_f = np.asarray(lat_f)
        lon_f = np.asarray(lon_f)
        dtype = np.dtype(dtype or common_dtype(u, v, lat_f, lon_f))
        f32 = np.dtype(np.float32)
        wind_f32 = dtype == np.dtype(np.float64) and common_dtype(u, v) == f32 and common_dtype(lat_f, lon_f) != f32
        if tuple(u.shape) != tuple(v.shape) or len(u.shape) != 3:
            raise ValueError("u and v must both be (time, latitude, longitude)")
        nt, ny_f, nx_f = (int(s) for s in u.shape)
        if lat_f.shape != (ny_f,) or lon_f.shape != (nx_f,):
            raise ValueError("coordinate lengths do not match the field")
        if not (np.all(np.diff(lat_f) > 0) and np.all(np.diff(lon_f) > 0)):
            raise ValueError("latitude and longitude must be ascending (sort first)")
        n = self.lib.lc_packed_elems(nt, ny_f, nx_f)
        self._use_current_stream()
        if wind_f32 and interp_order == 1 and lin_image is None:
            # float32 wind on float64 coordinates at order 1: the wind stays float32 -- its order-1 image as lc_field_pack builds
            # it for float32 fields, widened node by node inside the kernels (LC_F64_WIND_F32_LIN32: the bits of the float64
            # images, half the bytes, no conversion pass; LCS/trajectory.py:86-87,110-112 with SURVEY Q10).  The float64 planes
            # are made when a call needs them (_planes64: another order, the reference's outer-product clamp, sample()).
            u32, v32 = self.to_device(u, f32), self.to_device(v, f32)
            lin32 = self._empty((n,), f32)
            _capi.check(self.lib.lc_field_pack(self.ctx, self._ptr(u32), self._ptr(v32), _capi.LC_F32, nt, ny_f, nx_f, 1,
                                               self._ptr(lin32), None), self.lib)
            la, lo = lat_f.astype(dtype), lon_f.astype(dtype)
            return PackedField(None, None, None, nt, ny_f, nx_f, float(la[0]), float(la[-1]), float(lo[0]), float(lo[-1]), dtype,
                               True, 1, False, None, None, None, lin32, u32, v32, (u32._version, v32._version))
        if wind_f32 and interp_order == 3 and lin_image is None:
            # ... and at order 3 (the reference's default): scipy's spline coefficients of a float32 field are float64
            # (spline_filter(output=float64) inside map_coordinates), so the coefficient image is packed in float64 STRAIGHT from
            # the float32 planes (lc_field_pack(LC_F64_WIND_F32): no float64 copy of the wind), and the planes themselves are the
            # order-1 source of the pole rows.
            u32, v32 = self.to_device(u, f32), self.to_device(v, f32)
            cub = self._empty((n,), dtype)
            _capi.check(self.lib.lc_field_pack(self.ctx, self._ptr(u32), self._ptr(v32), _capi.LC_F64_WIND_F32, nt, ny_f, nx_f, 3,
                                               self._ptr(cub), None), self.lib)
            la, lo = lat_f.astype(dtype), lon_f.astype(dtype)
            return PackedField(None, cub, None, nt, ny_f, nx_f, float(la[0]), float(la[-1]), float(lo[0]), float(lo[-1]), dtype,
                               True, 3, False, None, None, None, None, u32, v32, (u32._version, v32._version))
        ud = self.to_device(u, dtype)
        vd = self.to_device(v, dtype)
        if fuse_levels is None:
            fuse_levels = True
        if wind_f32 or interp_order in (2, 4, 5):   # general orders: generic direct kernel, two-sample form
            fuse_levels = False
        ext = None
        if lin_image is None:
            lin_image = dtype == f32 and interp_order == 1
        fuse_raw = False
        if fuse_levels and nt >= 2 and dtype != f32 and interp_order == 1 and not lin_image:
            fuse_raw = not (self.EXT_IMAGE_F64 if ext_image is None else ext_image)
        if fuse_levels and nt >= 2 and dtype != f32 and interp_order == 3:
            fuse_raw = not (self.EXT_IMAGE_F64_O3 if ext_image is None else ext_image)
        if fuse_levels and nt >= 2 and not fuse_raw:
            ext = self._empty((self.lib.lc_packed_elems(nt - 1, ny_f, nx_f),), dtype)
        if dtype == f32 and interp_order == 1 and not lin_image:
            raise ValueError("float32 at interp_order=1 samples the order-1 image: lin_image cannot be False")
        lin = self._empty((n,), dtype) if lin_image else None
        if lin is not None or (interp_order == 1 and ext is not None):
            _capi.check(self.lib.lc_field_pack(self.ctx, self._ptr(ud), self._ptr(vd), _NP2LC[dtype], nt, ny_f, nx_f, 1,
                                               self._ptr(lin), self._ptr(ext if interp_order == 1 else None)), self.lib)
        cub = None
        if interp_order != 1:
            cub = self._empty((n,), dtype)
            _capi.check(self.lib.lc_field_pack(self.ctx, self._ptr(ud), self._ptr(vd), _NP2LC[dtype], nt, ny_f, nx_f,
                                               int(interp_order), self._ptr(cub), self._ptr(ext)), self.lib)
        # coordinate extremes in the arithmetic dtype (what .min()/.max() give numpy)
        la = lat_f.astype(dtype)
        lo = lon_f.astype(dtype)
        keep = lin is None
        return PackedField(lin, cub, ext, nt, ny_f, nx_f, float(la[0]), float(la[-1]), float(lo[0]), float(lo[-1]), dtype,
                           wind_f32, int(interp_order), fuse_raw, ud if keep else None, vd if keep else None,
                           (ud._version, vd._version) if keep else None)

    # ------------------------------------------------------------------ global pre-processing (LCS.py:105-118)
    def regrid(self, u, lat, lon, lats, lons):
        """``u.interp(linear)`` onto (lats, lons) with nearest fill outside the source range (LCS/LCS.py:107-114).
        u: (nt, nlat, nlon) array or device tensor.  Returns a float64 device tensor (nt, len(lats), len(lons))."""
        dtype = common_dtype(u)
        ud = self.to_device(u, dtype)
        nt, ny_s, nx_s = (int(s) for s in ud.shape)
        c = [np.ascontiguousarray(a, dtype=np.float64) for a in (lat, lon, lats, lons)]
        if c[0].size != ny_s or c[1].size != nx_s:
            raise ValueError("coordinate lengths do not match the field")
        out = self._empty((nt, c[2].size, c[3].size), np.float64)
        self._use_current_stream()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _capi.check(self.lib.lc_regrid_common_grid(self.ctx, self._ptr(ud), _NP2LC[dtype], nt, ny_s, nx_s, p(c[0]), p(c[1]),
                                                   p(c[2]), int(c[2].size), p(c[3]), int(c[3].size), self._ptr(out)), self.lib)
        return out

    def spectral_truncate(self, f, T=20, gridtype="regular"):
        """Triangular truncation at total wavenumber T of fields (..., nlat, nlon), latitude ascending, on
        SPHEREPACK's equally spaced grid or (``gridtype='gaussian'``) on Gaussian latitudes (LCS/LCS.py:115-118).
        Same shape and dtype back (device tensor)."""
        dtype = common_dtype(f)
        fd = self.to_device(f, dtype)
        shape = tuple(int(s) for s in fd.shape)
        nlat, nlon = shape[-2], shape[-1]
        nb = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        out = self._empty(shape, dtype)
        self._use_current_stream()
        gt = {"regular": _capi.LC_GRID_REGULAR, "gaussian": _capi.LC_GRID_GAUSSIAN}[gridtype]
        _capi.check(self.lib.lc_spectral_truncate(self.ctx, self._ptr(fd), _NP2LC[dtype], nb, nlat, nlon, int(T), gt,
                                                  self._ptr(out)), self.lib)
        return out

    # ------------------------------------------------------------------ K1
    def advect(self, field: PackedField, seed_lat, seed_lon, timestep, SETTLS_order=0, interp_order=1,
               cyclic_xboundary=True, t0=0, nsteps=None, return_traj=False, row0=0, ny_global=None, halo=None,
               noncyclic_clamp=None, start=None, out=None, global_rows=None):
        """Departure points of the seed rows given.  Returns (x, y[, traj_x, traj_y]) device tensors.

        ``global_rows`` (instead of ``row0``): the rows given are NOT one contiguous block of the global grid but an
        ascending selection of its rows (``sharded``'s interleaved chunks: several windows concatenated); entry i is the
        global row of ``seed_lat[i]``.  Advection is per seed (LCS/trajectory.py:80-126) and the only thing the kernels
        derive from a seed's global row is whether it is one of the ``interp_order`` rows next to either pole (Q3), so
        the call is expressed through ``row0`` / ``ny_global`` values that mark exactly those rows (:meth:`pole_window`);
        a selection that holds only part of a pole's rows, or holds them anywhere but at its own ends, is refused.

        ``out=(x, y)``: write the results into these ``(ny, nx)`` tensors (may be the ``start`` tensors: in place)
        instead of allocating.

        ``start=(x, y)``: continue from these ``(ny, nx)`` positions instead of the seed grid (``lc_advect_from``):
        levels [t0, t0+a) followed by [t0+a, t0+a+b) from the first call's result equals one call over a+b levels.

        ``noncyclic_clamp`` (only with ``cyclic_xboundary=False``): see :func:`x_boundary_mode`.

        ``halo=(n_lo, n_hi)``: return ``(n_lo + ny + n_hi, nx)`` buffers with the results in the middle
        rows, so a row-sharded caller can receive its neighbours' rows in place (sharded.py)."""
        if interp_order != 1 and field.order != interp_order:
            raise ValueError(f"field was prepared for interp_order={field.order}")
        dtype = field.dtype
        slat = self.to_device(seed_lat, dtype)
        slon = self.to_device(seed_lon, dtype)
        ny, nx = int(slat.numel()), int(slon.numel())
        ny_global = ny if ny_global is None else int(ny_global)
        whole = int(row0) == 0 and ny == ny_global
        if global_rows is not None:
            if int(row0) != 0 or halo:
                raise ValueError("global_rows replaces row0 (and carries its own halo rows)")
            row0, ny_global = self.pole_window(global_rows, ny, ny_global, int(interp_order))
            whole = False
        nsteps = field.nt - 1 - t0 if nsteps is None else int(nsteps)
        n_lo, n_hi = halo if halo else (0, 0)
        if out is not None:
            if halo:
                raise ValueError("out= and halo= cannot be combined")
            x_buf, y_buf = out
            want = getattr(self.torch, np.dtype(dtype).name)
            for b in (x_buf, y_buf):
                if tuple(b.shape) != (ny, nx) or b.dtype != want or not b.is_contiguous() or b.device != self.device:
                    raise ValueError(f"out tensors must be contiguous ({ny}, {nx}) {np.dtype(dtype).name} tensors on {self.device}")
        else:
            x_buf = self._empty((n_lo + ny + n_hi, nx), dtype)
            y_buf = self._empty((n_lo + ny + n_hi, nx), dtype)
        x, y = x_buf[n_lo:n_lo + ny], y_buf[n_lo:n_lo + ny]
        if halo:  # rows to be received: NaN until the exchange fills them, so a skipped exchange cannot pass unnoticed
            for buf in (x_buf, y_buf):
                buf[:n_lo].fill_(float("nan"))
                buf[n_lo + ny:].fill_(float("nan"))
        tx = ty = None
        if return_traj:
            tx = self._empty((nsteps + 1, ny, nx), dtype)
            ty = self._empty((nsteps + 1, ny, nx), dtype)
        sx = sy = None
        if start is not None:
            sx, sy = (self.to_device(a, dtype) for a in start)
            if tuple(sx.shape) != (ny, nx) or tuple(sy.shape) != (ny, nx):
                raise ValueError(f"start positions must be two ({ny}, {nx}) arrays")
        self._use_current_stream()
        a = self._advect_args(field, interp_order, slat, ny, slon, nx, row0, ny_global, sx, sy, timestep, SETTLS_order,
                              x_boundary_mode(cyclic_xboundary, noncyclic_clamp, whole),
                              t0, nsteps, 1, 0, x, y, tx, ty)
        _capi.check(self.lib.lc_advect_ex(self.ctx, C.byref(a)), self.lib)
        if halo:
            x, y = x_buf, y_buf
        return (x, y, tx, ty) if return_traj else (x, y)

    @staticmethod
    def pole_window(global_rows, ny: int, ny_global: int, order: int):
        """``(row0, ny_global)`` to hand ``lc_advect`` for an ascending SELECTION of the global grid's rows, such that the
        kernels' pole rule -- local row i is a pole row iff ``row0 + i < order`` or ``row0 + i >= ny_global - order`` (Q3: the
        ``order`` rows next to either pole take the order-1 / 'constant' sample) -- marks exactly the selected rows that are
        pole rows of the real grid.  The selection must hold a pole's rows completely and as its own first / last rows, or
        not at all (``sharded.interleaved_chunks`` does: the first chunk starts at row 0, the last one ends at the last row)."""
        g = np.asarray(global_rows, dtype=np.int64)
        if g.shape != (ny,) or (ny > 1 and not np.all(np.diff(g) > 0)) or g[0] < 0 or g[-1] >= ny_global:
            raise ValueError(f"global_rows: {ny} ascending row indices inside [0, {ny_global})")
        is_pole = (g < order) | (g >= ny_global - order)
        row0 = 0 if g[0] < order else order
        nyg = row0 + ny + (0 if g[-1] >= ny_global - order else order)
        i = np.arange(ny) + row0
        if not np.array_equal((i < order) | (i >= nyg - order), is_pole):
            raise ValueError("global_rows: the selection holds part of a pole's rows, or holds them elsewhere than at its ends")
        return int(row0), int(nyg)

    def advect_batch(self, field: PackedField, seed_lat, seed_lon, timestep, n_members: int, nsteps: int, SETTLS_order=0,
                     interp_order=1, cyclic_xboundary=True, t0=0, t0_stride=1, start=None, out=None):
        """Departure points of an ENSEMBLE of start times over one seed grid (``lc_advect_batch``): member ``m`` runs
        ``nsteps`` steps from time level ``t0 + m * t0_stride``.  Returns ``(x, y)`` of shape ``(n_members, ny, nx)``.
        One launch per level chunk covers every member (:meth:`set_level_chunk`); each member's result equals
        ``advect(t0=t0 + m * t0_stride, nsteps=nsteps)`` bit for bit.  ``start`` / ``out``: ``(n_members, ny, nx)``
        tensors to continue from / write into (may be the same)."""
        if interp_order != 1 and field.order != interp_order:
            raise ValueError(f"field was prepared for interp_order={field.order}")
        if not cyclic_xboundary:
            raise ValueError("advect_batch: the reference's non-cyclic clamp is decided per member; call advect for each")
        dtype = field.dtype
        slat, slon = self.to_device(seed_lat, dtype), self.to_device(seed_lon, dtype)
        ny, nx, n = int(slat.numel()), int(slon.numel()), int(n_members)
        want = getattr(self.torch, np.dtype(dtype).name)

        def chk(t, what):
            if tuple(t.shape) != (n, ny, nx) or t.dtype != want or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{what} tensors must be contiguous ({n}, {ny}, {nx}) {np.dtype(dtype).name} tensors on {self.device}")
            return t
        x, y = (chk(t, "out") for t in out) if out is not None else (self._empty((n, ny, nx), dtype), self._empty((n, ny, nx), dtype))
        sx, sy = (chk(t, "start") for t in start) if start is not None else (None, None)
        self._use_current_stream()
        a = self._advect_args(field, interp_order, slat, ny, slon, nx, 0, ny, sx, sy, timestep, SETTLS_order,
                              _capi.LC_X_CYCLIC, t0, nsteps, n, t0_stride, x, y, None, None)
        _capi.check(self.lib.lc_advect_ex(self.ctx, C.byref(a)), self.lib)
        return x, y

    def _advect_args(self, field, interp_order, slat, ny, slon, nx, row0, ny_global, sx, sy, timestep, K, xmode, t0, nsteps,
                     n_members, t0_stride, x, y, tx, ty) -> "_capi.AdvectArgs":
        """``lc_advect_args`` of one call: the field's images, and its raw planes as the order-1 source where it has no
        lin image."""
        p = lambda t: t.data_ptr() if t is not None else None
        self._check_planes(field)
        if field.u32 is not None and field.lin32 is None and interp_order == 3 == field.order and xmode != _capi.LC_X_CLAMP_REFERENCE_OUTER:
            # float32 wind on float64 coordinates at order 3: float64 coefficients, the float32 planes for the pole rows
            return _capi.AdvectArgs(
                struct_size=C.sizeof(_capi.AdvectArgs), packed_cub=p(field.cub), u_raw=p(field.u32), v_raw=p(field.v32),
                dtype=_capi.LC_F64_WIND_F32_LIN32, nt=field.nt, ny_f=field.ny_f, nx_f=field.nx_f, lat_min=field.lat_min,
                lat_max=field.lat_max, lon_min=field.lon_min, lon_max=field.lon_max, seed_lat_dev=p(slat), ny=int(ny),
                seed_lon_dev=p(slon), nx=int(nx), row0=int(row0), ny_global=int(ny_global), x_start=p(sx), y_start=p(sy),
                timestep=float(timestep), settls_order=int(K), interp_order=3, cyclic_x=int(xmode), t0=int(t0), nsteps=int(nsteps),
                n_members=int(n_members), t0_stride=int(t0_stride), x_out=p(x), y_out=p(y), traj_x=p(tx), traj_y=p(ty), fuse_levels_raw=0)
        if field.lin32 is not None and interp_order == 1 and xmode != _capi.LC_X_CLAMP_REFERENCE_OUTER:
            # float32 wind on float64 coordinates, the wind kept float32 (prepare_field)
            return _capi.AdvectArgs(
                struct_size=C.sizeof(_capi.AdvectArgs), packed_lin=p(field.lin32), dtype=_capi.LC_F64_WIND_F32_LIN32, nt=field.nt,
                ny_f=field.ny_f, nx_f=field.nx_f, lat_min=field.lat_min, lat_max=field.lat_max, lon_min=field.lon_min,
                lon_max=field.lon_max, seed_lat_dev=p(slat), ny=int(ny), seed_lon_dev=p(slon), nx=int(nx), row0=int(row0),
                ny_global=int(ny_global), x_start=p(sx), y_start=p(sy), timestep=float(timestep), settls_order=int(K),
                interp_order=1, cyclic_x=int(xmode), t0=int(t0), nsteps=int(nsteps), n_members=int(n_members),
                t0_stride=int(t0_stride), x_out=p(x), y_out=p(y), traj_x=p(tx), traj_y=p(ty), fuse_levels_raw=0)
        dt = _capi.LC_F64_WIND_F32 if field.wind_f32 else _NP2LC[field.dtype]
        self._planes64(field)
        self._ensure_lin(field, interp_order)
        return _capi.AdvectArgs(
            struct_size=C.sizeof(_capi.AdvectArgs), packed_lin=p(field.lin),
            packed_cub=p(field.cub if interp_order != 1 else None),
            packed_ext=p(field.ext if field.order == interp_order else None), u_raw=p(field.u), v_raw=p(field.v),
            dtype=dt, nt=field.nt, ny_f=field.ny_f, nx_f=field.nx_f, lat_min=field.lat_min, lat_max=field.lat_max,
            lon_min=field.lon_min, lon_max=field.lon_max, seed_lat_dev=p(slat), ny=int(ny), seed_lon_dev=p(slon), nx=int(nx),
            row0=int(row0), ny_global=int(ny_global), x_start=p(sx), y_start=p(sy), timestep=float(timestep),
            settls_order=int(K), interp_order=int(interp_order), cyclic_x=int(xmode), t0=int(t0), nsteps=int(nsteps),
            n_members=int(n_members), t0_stride=int(t0_stride), x_out=p(x), y_out=p(y), traj_x=p(tx), traj_y=p(ty),
            fuse_levels_raw=int(bool(field.fuse_raw and interp_order == field.order)))

    def _planes64(self, field: PackedField):
        """A wind_f32 field prepared at order 1 keeps its wind float32; the float64 planes (the order-1 source of every other
        call form: lc_advect's LC_F64_WIND_F32 with the reference's outer-product clamp, :meth:`sample`) are made here, once."""
        if field.u is None and field.u32 is not None:
            field.u, field.v = self.to_device(field.u32, field.dtype), self.to_device(field.v32, field.dtype)
            field.planes_version = (field.u._version, field.v._version)

    @staticmethod
    def _check_planes(field: PackedField):
        """The borrowed wind planes (``u`` / ``v``, and the float32 ``u32`` / ``v32`` of a float32 wind on float64 coordinates)
        must not have been written in place since ``prepare_field``: the kernels read them live (pole rows, Euler samples)
        next to images packed from their old values.  Called on EVERY path that builds a call's arguments."""
        stale = (field.u is not None and field.planes_version is not None
                 and (field.u._version, field.v._version) != field.planes_version) or \
                (field.u32 is not None and field.planes32_version is not None
                 and (field.u32._version, field.v32._version) != field.planes32_version)
        if stale:
            raise RuntimeError("the wind tensors given to prepare_field were modified in place afterwards: the field's packed "
                               "images no longer match them (prepare the field again, or pass copies)")

    def _ensure_lin(self, field: PackedField, interp_order: int):
        """The order-1 source of a call on ``field``, checked and -- where it is an image that does not exist yet -- built.

        A float32 field prepared for another order and now used at order 1 ("order 1 is always available"): its kernels
        read the order-1 image's 16-byte node pairs, so the image is packed here, once, and kept on the field (``advect``
        and ``sample`` alike).  Everywhere else the raw planes ``field.u`` / ``field.v`` are the order-1 source; they are
        BORROWED from the caller when ``prepare_field`` was handed device tensors, so an in-place write to them since
        then (a time loop refilling its buffers) would silently change what the pole rows and the float64 Euler sample
        read while the packed images still hold the old wind: refused here by the tensors' version counters."""
        self._check_planes(field)
        if field.lin is None and field.dtype == np.dtype(np.float32) and interp_order == 1:
            field.lin = self._empty((self.lib.lc_packed_elems(field.nt, field.ny_f, field.nx_f),), field.dtype)
            self._use_current_stream()
            _capi.check(self.lib.lc_field_pack(self.ctx, self._ptr(field.u), self._ptr(field.v), _NP2LC[field.dtype], field.nt,
                                               field.ny_f, field.nx_f, 1, self._ptr(field.lin), None), self.lib)

    def sample(self, field: PackedField, pos_x, pos_y, level=0, interp_order=1, row0=0, ny_global=None):
        """tools.xr_map_coordinates for (u, v) of one time level at positions (ny, nx) in degrees."""
        if interp_order != 1 and field.order != interp_order:
            raise ValueError(f"field was prepared for interp_order={field.order}")
        self._planes64(field)
        self._ensure_lin(field, interp_order)
        dtype = field.dtype
        px = self.to_device(pos_x, dtype)
        py = self.to_device(pos_y, dtype)
        ny, nx = (int(s) for s in px.shape)
        ny_global = ny if ny_global is None else int(ny_global)
        ou = self._empty((ny, nx), dtype)
        ov = self._empty((ny, nx), dtype)
        self._use_current_stream()
        _capi.check(self.lib.lc_sample_raw(
            self.ctx, self._ptr(field.lin), self._ptr(field.cub if interp_order != 1 else None), self._ptr(field.u),
            self._ptr(field.v), _NP2LC[dtype],
            field.nt, field.ny_f, field.nx_f, field.lat_min, field.lat_max, field.lon_min, field.lon_max, int(level),
            self._ptr(px), self._ptr(py), ny, nx, int(row0), ny_global, int(interp_order), self._ptr(ou),
            self._ptr(ov)), self.lib)
        return ou, ov

    # ------------------------------------------------------------------ K3
    def sigma(self, x_dep, y_dep, seed_lat_rows, dlat, dlon, ny_global=None, in_row0=0, out_row0=None,
              n_out_rows=None, fd_fp32_cast=True, tensor_layout="reference"):
        """sigma_max for rows [out_row0, out_row0+n_out_rows) of a global grid, from the
        departure rows [in_row0, in_row0+x_dep.shape[0]) (which must include the 2-row halo)."""
        torch = self.torch
        dtype = np.dtype(str(x_dep.dtype).replace("torch.", "")) if isinstance(x_dep, torch.Tensor) \
            else common_dtype(x_dep, y_dep)
        xd = self.to_device(x_dep, dtype)
        yd = self.to_device(y_dep, dtype)
        n_in, nx = (int(s) for s in xd.shape)
        slat = self.to_device(seed_lat_rows, dtype)
        if slat.numel() != n_in:
            raise ValueError("seed_lat_rows must have one latitude per input row")
        ny_global = n_in if ny_global is None else int(ny_global)
        out_row0 = in_row0 if out_row0 is None else int(out_row0)
        n_out_rows = n_in if n_out_rows is None else int(n_out_rows)
        sig = self._empty((n_out_rows, nx), dtype)
        self._use_current_stream()
        _capi.check(self.lib.lc_sigma(self.ctx, self._ptr(xd), self._ptr(yd), _NP2LC[dtype], int(in_row0), n_in, nx,
                                      ny_global, self._ptr(slat), float(dlat), float(dlon), int(bool(fd_fp32_cast)),
                                      _LAYOUTS[tensor_layout], out_row0, n_out_rows, self._ptr(sig)), self.lib)
        return sig

    def flowmap_gradient(self, x_dep, y_dep, seed_lat, dlat, dlon, fd_fp32_cast=True):
        """The (9, ny, nx) def_tensor of LCS.flowmap_gradient (LCS/LCS.py:195-223) as a device tensor."""
        torch = self.torch
        dtype = np.dtype(str(x_dep.dtype).replace("torch.", "")) if isinstance(x_dep, torch.Tensor) \
            else common_dtype(x_dep, y_dep)
        xd = self.to_device(x_dep, dtype)
        yd = self.to_device(y_dep, dtype)
        ny, nx = (int(s) for s in xd.shape)
        slat = self.to_device(seed_lat, dtype)
        out = self._empty((9, ny, nx), dtype)
        self._use_current_stream()
        _capi.check(self.lib.lc_flowmap_gradient(self.ctx, self._ptr(xd), self._ptr(yd), _NP2LC[dtype], ny, nx,
                                                 self._ptr(slat), float(dlat), float(dlon), int(bool(fd_fp32_cast)),
                                                 self._ptr(out)), self.lib)
        return out

    def index_derivative(self, a, dim, isglobal=True):
        """tools.fourth_order_derivative(a, dim, isglobal) (LCS/tools.py:190-245); dtype preserved."""
        torch = self.torch
        dtype = np.dtype(str(a.dtype).replace("torch.", ""))
        if dtype not in _NP2LC:
            raise ValueError(f"dtype {dtype} unsupported (float32 / float64)")
        ad = self.to_device(a, dtype)
        ny, nx = (int(s) for s in ad.shape)
        out = self._empty((ny, nx), dtype)
        self._use_current_stream()
        _capi.check(self.lib.lc_fourth_order_derivative(self.ctx, self._ptr(ad), _NP2LC[dtype], ny, nx, int(dim),
                                                        int(bool(isglobal)), self._ptr(out)), self.lib)
        return out

    def ridge_classify(self, hxx, hxy, hyy, gx, gy, tolerance, return_eigvec=False):
        """Per-point step of tools.find_ridges_spherical_hessian (LCS/tools.py:99-138).
        Returns (mask, eigmin, dt) float64 device tensors shaped like the inputs, plus -- with
        ``return_eigvec`` -- the reference's row-indexed eigenvector as a ``(2, *shape)`` tensor."""
        t = [self.to_device(a, np.float64) for a in (hxx, hxy, hyy, gx, gy)]
        shape = tuple(t[0].shape)
        n = int(t[0].numel())
        mask, eigmin, dt = (self._empty(shape, np.float64) for _ in range(3))
        vec = self._empty((2,) + shape, np.float64) if return_eigvec else None
        self._use_current_stream()
        _capi.check(self.lib.lc_ridge_classify(self.ctx, *(self._ptr(a) for a in t), n, float(tolerance),
                                               self._ptr(mask), self._ptr(eigmin), self._ptr(dt),
                                               self._ptr(vec) if return_eigvec else None), self.lib)
        return (mask, eigmin, dt, vec) if return_eigvec else (mask, eigmin, dt)

    # ------------------------------------------------------------------ multi-GPU (RCCL through the C ABI)
    COMM_ID_BYTES = 128

    def comm_unique_id(self) -> bytes:
        """The id rank 0 creates and every rank passes to :meth:`comm_create` (``lc_comm_unique_id``)."""
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        _capi.check(self.lib.lc_comm_unique_id(buf, self.COMM_ID_BYTES), self.lib)
        return buf.raw

    def comm_create(self, nranks: int, rank: int, unique_id: bytes):
        """RCCL communicator over ``nranks`` processes, this one on this engine's GPU (collective call)."""
        comm = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), self.COMM_ID_BYTES)
        _capi.check(self.lib.lc_comm_create(self.ctx, int(nranks), int(rank), buf, self.COMM_ID_BYTES, C.byref(comm)),
                    self.lib)
        return comm

    def comm_count(self, comm):
        """(nranks, rank) as RCCL itself reports them for the communicator (``lc_comm_count``)."""
        n, r = C.c_int(), C.c_int()
        _capi.check(self.lib.lc_comm_count(comm, C.byref(n), C.byref(r)), self.lib)
        return n.value, r.value

    def comm_destroy(self, comm):
        if comm:
            self.lib.lc_comm_destroy(comm)

    def halo_exchange(self, comm, x_ext, y_ext, n_lo: int, n_hi: int):
        """In-place 2-row halo exchange of the departure points with the previous / next rank
        (``lc_halo_exchange``: RCCL send/recv on the current stream, no staging copies)."""
        if x_ext.shape != y_ext.shape or x_ext.dtype != y_ext.dtype or not (x_ext.is_contiguous() and y_ext.is_contiguous()):
            raise ValueError("x_ext and y_ext must be contiguous (rows, nx) tensors of one dtype")
        dtype = np.dtype(str(x_ext.dtype).replace("torch.", ""))
        self._use_current_stream()
        _capi.check(self.lib.lc_halo_exchange(self.ctx, comm, self._ptr(x_ext), self._ptr(y_ext), _NP2LC[dtype],
                                              int(x_ext.shape[0]), int(x_ext.shape[1]), int(n_lo), int(n_hi)), self.lib)

    def gaussian_filter(self, a, sigma):
        """scipy.ndimage.gaussian_filter(a, sigma) on the device (LCS/LCS.py:187-190)."""
        torch = self.torch
        dtype = np.dtype(str(a.dtype).replace("torch.", "")) if isinstance(a, torch.Tensor) else common_dtype(a)
        ad = self.to_device(a, dtype)
        ny, nx = (int(s) for s in ad.shape)
        tmp = self._empty((ny, nx), dtype)
        out = self._empty((ny, nx), dtype)
        self._use_current_stream()
        _capi.check(self.lib.lc_gaussian_filter(self.ctx, self._ptr(ad), _NP2LC[dtype], ny, nx, float(sigma),
                                                self._ptr(tmp), self._ptr(out)), self.lib)
        return out

    # ------------------------------------------------------------------ pack and advect, pipelined
    PIPELINE_CHUNK = 24      # time levels per pack / advect stage of the pipelined form (measured: profiles/r04/pipelined_pack_ab.txt)

    def pipeline_pays(self, dtype, interp_order, fuse_levels, nsteps, n_seeds, cyclic_xboundary, return_traj=False) -> bool:
        """Where packing chunk k+1 on a side stream while chunk k is advected is the default: nowhere since round 5.  Until
        then float64 at order 3 in the fused-level form took it (the pack's two prefilter sweeps ran at half the HBM rate
        next to a latency-bound advect kernel: 10.9 -> 9.75 ms per step on BASELINE configs[1]); with both sweeps in one
        pass (``prefilter_fused_stream_kernel``: pack 4.6 -> 2.95 ms) the serial form takes 9.0 ms and the pipelined one
        9.1-9.5 at any chunk length (``profiles/r05/c2_o3_fused_prefilter_ab.txt``): the two kernels now want the same
        CUs.  float32 (its advect kernels saturate the VALU) and float64 at order 1 never gained from it.  The pipelined
        form stays available (``pack_and_advect(pipeline=True)``, bit-identical)."""
        return False

    def pack_and_advect(self, u, v, lat_f, lon_f, seed_lat, seed_lon, timestep, SETTLS_order=0, interp_order=1,
                        cyclic_xboundary=True, fuse_levels=None, pipeline=None, chunk=None, return_traj=False,
                        noncyclic_clamp=None, ext_image=None):
        """:meth:`prepare_field` + :meth:`advect` of the whole series in one call.  Returns ``(field, x, y[, traj_x, traj_y])``.

        ``pipeline`` (None: where it pays, :meth:`pipeline_pays`): the series is cut into chunks of ``chunk`` time levels;
        the images of chunk k+1 are packed on a side HIP stream while the advect kernel works through chunk k on the
        current stream (an event per chunk), each advect continuing in place from the previous one (``lc_advect_from``:
        bit-identical to one call, LCS/trajectory.py:80-126 carries only positions from level to level).  The level shared
        by two chunks is packed by both (same values; the earlier chunk's advect never reads it).  Results are
        bit-identical to the serial form; the field returned is reusable (with ``SETTLS_order=0`` it holds no fused-level image)."""
        torch = self.torch
        lat_f, lon_f = np.asarray(lat_f), np.asarray(lon_f)
        dtype = common_dtype(u, v, lat_f, lon_f)
        if fuse_levels is None:
            fuse_levels = True
        if int(SETTLS_order) == 0 and fuse_levels and ext_image is None:
            # SETTLS_order = 0 (the library default, LCS/trajectory.py:14) takes one Euler sample per level and never reads the
            # fused-level image: it is not built.  float64: the kernels' no-image forms (the same Euler sample, bit for bit --
            # at order 1 no packed image at all, at order 3 the coefficients only); float32: the order-1 / coefficient image only.
            # configs[1] at K = 0: pack 1.29 -> 0 ms (order 1), 3.17 -> 1.85 ms (order 3).  The field returned holds what THIS
            # call needed: advected again with SETTLS_order > 0 it takes the two-sample form.
            if dtype != np.dtype(np.float32):
                ext_image = False
            else:
                fuse_levels = False
        nt = int(u.shape[0])
        ny, nx = len(seed_lat), len(seed_lon)
        if pipeline is None:
            pipeline = self.pipeline_pays(dtype, interp_order, fuse_levels, nt - 1, ny * nx, cyclic_xboundary, return_traj)
        chunk = self.PIPELINE_CHUNK if chunk is None else int(chunk)
        f32 = np.dtype(np.float32)
        wind_f32 = dtype != f32 and common_dtype(u, v) == f32
        can = (cyclic_xboundary and not return_traj and interp_order in (1, 3) and fuse_levels and not wind_f32 and nt - 1 > chunk >= 1
               and not (dtype == f32 and interp_order == 1))      # (float32 at order 1 carries a lin image too: serial form)
        if not (pipeline and can):
            field = self.prepare_field(u, v, lat_f, lon_f, interp_order, fuse_levels=fuse_levels, ext_image=ext_image)
            res = self.advect(field, seed_lat, seed_lon, timestep, SETTLS_order, interp_order, cyclic_xboundary,
                              return_traj=return_traj, noncyclic_clamp=noncyclic_clamp)
            return (field, *res)
        if tuple(u.shape) != tuple(v.shape) or len(u.shape) != 3 or lat_f.shape != (u.shape[1],) or lon_f.shape != (u.shape[2],):
            raise ValueError("u and v must both be (time, latitude, longitude) with matching coordinates")
        if not (np.all(np.diff(lat_f) > 0) and np.all(np.diff(lon_f) > 0)):
            raise ValueError("latitude and longitude must be ascending (sort first)")
        ny_f, nx_f = int(u.shape[1]), int(u.shape[2])
        ud, vd = self.to_device(u, dtype), self.to_device(v, dtype)
        le = self.lib.lc_packed_elems(1, ny_f, nx_f)
        cub = self._empty((le * nt,), dtype) if interp_order == 3 else None
        no_ext = dtype != f32 and interp_order == 3 and not (self.EXT_IMAGE_F64_O3 if ext_image is None else ext_image)   # the kernels form it from cub (prepare_field)
        ext = None if no_ext else self._empty((le * (nt - 1),), dtype)
        la, lo = lat_f.astype(dtype), lon_f.astype(dtype)
        field = PackedField(None, cub, ext, nt, ny_f, nx_f, float(la[0]), float(la[-1]), float(lo[0]), float(lo[-1]), dtype,
                            False, int(interp_order), no_ext, ud, vd, (ud._version, vd._version))
        slat, slon = self.to_device(seed_lat, dtype), self.to_device(seed_lon, dtype)
        x, y = self._empty((ny, nx), dtype), self._empty((ny, nx), dtype)
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(self.device)
        side = self._side_stream
        side.wait_stream(cur)                  # the wind, the seeds and the buffers above belong to the current stream
        # (no record_stream on the images: every pack on the side stream is awaited by the current stream below, so by the
        #  time the field can be freed -- in the current stream's order -- the side stream is done with it; recording the
        #  6.8 GB of images instead made the caching allocator hold the blocks back and cudaMalloc new ones every step)
        events, starts = [], list(range(0, nt - 1, chunk))
        try:
            with torch.cuda.stream(side):
                self._use_current_stream()
                for t0 in starts:
                    n = min(chunk, nt - 1 - t0)         # image levels [t0, t0 + n], ext levels [t0, t0 + n)
                    # Without the ext image the advect of chunk k itself reads level t0 + n (as the "next" level of its last
                    # step), so chunk k+1 must NOT pack that level again behind its back (the sweeps work in place: a reader
                    # would see half-filtered values): each level is packed once, by the chunk that first needs it.
                    l0 = t0 + 1 if (no_ext and t0 > 0) else t0
                    _capi.check(self.lib.lc_field_pack(
                        self.ctx, C.c_void_p(ud[l0:].data_ptr()), C.c_void_p(vd[l0:].data_ptr()), _NP2LC[dtype], t0 + n + 1 - l0, ny_f, nx_f,
                        int(interp_order), C.c_void_p(cub[le * l0:].data_ptr()) if cub is not None else None,
                        C.c_void_p(ext[le * t0:].data_ptr()) if ext is not None else None), self.lib)
                    e = torch.cuda.Event()
                    e.record(side)
                    events.append(e)
            for e, t0 in zip(events, starts):
                cur.wait_event(e)
                n = min(chunk, nt - 1 - t0)
                self.advect(field, slat, slon, timestep, SETTLS_order, interp_order, True, t0=t0, nsteps=n,
                            start=(x, y) if t0 else None, out=(x, y))
        finally:
            # whatever happened above -- a pack refused half way through the loop included -- nothing of this call is left
            # running behind the current stream: the images were allocated on it and deliberately not record_stream'd
            cur.wait_stream(side)
        return field, x, y

    # ------------------------------------------------------------------ whole path
    def lcs_wind(self, u, v, lat_f, lon_f, seed_lat, seed_lon, timestep, SETTLS_order=0, interp_order=1, cyclic_xboundary=True,
                 fuse_levels=None, gauss_sigma=None, fd_fp32_cast=True, tensor_layout="reference", return_traj=False,
                 noncyclic_clamp=None, pipeline=None):
        """:meth:`pack_and_advect` -> (smooth) -> sigma: the whole path from the raw wind series (what the drop-in surface
        calls).  Returns the dict of :meth:`lcs` plus ``"field"``."""
        dtype = common_dtype(u, v, np.asarray(lat_f), np.asarray(lon_f))
        seed_lat, seed_lon = np.asarray(seed_lat, dtype=dtype), np.asarray(seed_lon, dtype=dtype)
        res = self.pack_and_advect(u, v, lat_f, lon_f, seed_lat, seed_lon, timestep, SETTLS_order, interp_order,
                                   cyclic_xboundary, fuse_levels, pipeline, None, return_traj, noncyclic_clamp)
        out = self._sigma_of(res[1], res[2], seed_lat, seed_lon, gauss_sigma, fd_fp32_cast, tensor_layout)
        out["field"] = res[0]
        if return_traj:
            out["traj_x"], out["traj_y"] = res[3], res[4]
        return out

    def _sigma_of(self, x, y, seed_lat, seed_lon, gauss_sigma, fd_fp32_cast, tensor_layout):
        xs, ys = x, y
        # scipy's gaussian_filter returns an unsmoothed copy for sigma = 0 (LCS/LCS.py:187-190): skip the filter
        if isinstance(gauss_sigma, (float, int)) and not isinstance(gauss_sigma, bool) and gauss_sigma > 1e-15:
            xs = self.gaussian_filter(x, gauss_sigma)
            ys = self.gaussian_filter(y, gauss_sigma)
        # spacing evaluated in the coordinate dtype, as lat[1]-lat[0] is in numpy (tools.py:255-256)
        dlat = float(seed_lat[1] - seed_lat[0])
        dlon = float(seed_lon[1] - seed_lon[0])
        sig = self.sigma(xs, ys, seed_lat, dlat, dlon, fd_fp32_cast=fd_fp32_cast, tensor_layout=tensor_layout)
        return {"sigma": sig, "x_dep": x, "y_dep": y}

    def lcs(self, field: PackedField, seed_lat, seed_lon, timestep, SETTLS_order=0, interp_order=1,
            cyclic_xboundary=True, t0=0, nsteps=None, gauss_sigma=None, fd_fp32_cast=True,
            tensor_layout="reference", return_traj=False, noncyclic_clamp=None):
        """advect -> (smooth) -> sigma on one GPU.  Returns dict of device tensors."""
        dtype = field.dtype
        seed_lat = np.asarray(seed_lat, dtype=dtype)
        seed_lon = np.asarray(seed_lon, dtype=dtype)
        res = self.advect(field, seed_lat, seed_lon, timestep, SETTLS_order, interp_order, cyclic_xboundary, t0,
                          nsteps, return_traj, noncyclic_clamp=noncyclic_clamp)
        out = self._sigma_of(res[0], res[1], seed_lat, seed_lon, gauss_sigma, fd_fp32_cast, tensor_layout)
        if return_traj:
            out["traj_x"], out["traj_y"] = res[2], res[3]
        return out

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


# ---------------------------------------------------------------------------
# torch-free route: host arrays through lc_lcs_host
# ---------------------------------------------------------------------------
def lcs_host(u, v, lat_f, lon_f, timestep, SETTLS_order=0, interp_order=3, cyclic_xboundary=False,
             seed_lat=None, seed_lon=None, t0=0, nsteps=None, gauss_sigma=None, fd_fp32_cast=True,
             tensor_layout="reference", return_traj=False, want_sigma=True, device=0, noncyclic_clamp=None,
             float64_fidelity=None, pipeline=None):
    """numpy in, numpy out, via the one-call C entry point.  Returns a dict.
    ``float64_fidelity``: ``'auto'`` (default) / ``'exact'`` / ``'fast'``, see :meth:`Engine.set_f64_fidelity`.
    ``pipeline`` (default on): the staged, level-chunk pipelined transfers of ``lc_ctx_set_host_pipeline``; ``False`` = the
    serial form (plain copies of the whole series, then the kernels).  Results are bit-identical."""
    lib = _capi.load()
    dtype = common_dtype(u, v, lat_f, lon_f, seed_lat, seed_lon)
    u = np.ascontiguousarray(u, dtype=dtype)
    v = np.ascontiguousarray(v, dtype=dtype)
    if u.shape != v.shape or u.ndim != 3:
        raise ValueError("u and v must both be (time, latitude, longitude)")
    lat_f = np.ascontiguousarray(lat_f, dtype=dtype)
    lon_f = np.ascontiguousarray(lon_f, dtype=dtype)
    seed_lat = lat_f if seed_lat is None else np.ascontiguousarray(seed_lat, dtype=dtype)
    seed_lon = lon_f if seed_lon is None else np.ascontiguousarray(seed_lon, dtype=dtype)
    nt, ny_f, nx_f = u.shape
    ny, nx = seed_lat.size, seed_lon.size
    nsteps = nt - 1 - t0 if nsteps is None else int(nsteps)
    out = {"x_dep": np.empty((ny, nx), dtype), "y_dep": np.empty((ny, nx), dtype)}
    if want_sigma:
        out["sigma"] = np.empty((ny, nx), dtype)
    if return_traj:
        out["traj_x"] = np.empty((nsteps + 1, ny, nx), dtype)
        out["traj_y"] = np.empty((nsteps + 1, ny, nx), dtype)

    def p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)

    ctx = _host_ctx(lib, device)
    with _HOST_LOCK:
        _capi.check(lib.lc_ctx_set_f64_fidelity(ctx, Engine._FIDELITY[float64_fidelity or "auto"]), lib)
        on = (os.environ.get("LCS_HOST_PIPELINE", "1")[:1] != "0") if pipeline is None else bool(pipeline)
        _capi.check(lib.lc_ctx_set_host_pipeline(ctx, int(on)), lib)
        gs = float(gauss_sigma) if isinstance(gauss_sigma, (float, int)) and not isinstance(gauss_sigma, bool) else 0.0
        _capi.check(lib.lc_lcs_host(
            ctx, p(u), p(v), _NP2LC[dtype], nt, ny_f, nx_f, p(lat_f), p(lon_f), p(seed_lat), ny, p(seed_lon), nx,
            float(timestep), int(SETTLS_order), int(interp_order), x_boundary_mode(cyclic_xboundary, noncyclic_clamp),
            int(t0), nsteps, gs,
            int(bool(fd_fp32_cast)), _LAYOUTS[tensor_layout], p(out.get("sigma")), p(out["x_dep"]), p(out["y_dep"]),
            p(out.get("traj_x")), p(out.get("traj_y"))), lib)
        marks = (C.c_double * 4)()
        lib.lc_ctx_last_host_marks(ctx, marks)
    out["host_marks_ms"] = {"buffers_allocated": marks[0], "uploads_and_launches_issued": marks[1], "kernels_done": marks[2],
                            "results_down": marks[3]}
    return out


# The one-call routes' context per device, created on first use and kept for the life of the process: it owns the pinned
# staging ring and the copy threads of lc_lcs_host (lc_ctx_set_host_pipeline), which cost far more to set up than a call
# takes.  Calls through it are serialised (a context is not re-entrant).
_HOST_CTX = {}
_HOST_LOCK = __import__("threading").Lock()


def _host_ctx(lib, device):
    with _HOST_LOCK:
        ctx = _HOST_CTX.get(int(device))
        if ctx is None:
            ctx = C.c_void_p()
            _capi.check(lib.lc_ctx_create(int(device), C.byref(ctx)), lib)
            _HOST_CTX[int(device)] = ctx
            import atexit
            atexit.register(lambda c=ctx: lib.lc_ctx_destroy(c))
        return ctx


def lcs_global_host(u, v, lat_f, lon_f, timestep, SETTLS_order=0, interp_order=3, interp_to_common_grid=True,
                    truncation=20, gauss_sigma=None, fd_fp32_cast=True, tensor_layout="reference", device=0,
                    float64_fidelity=None):
    """The reference's default global call form ``LCS(...)(ds, isglobal=True)`` (LCS/LCS.py:105-157) on numpy
    arrays, torch-free, through ``lc_lcs_global_host``: regrid to the common 0.5 degree grid, T-truncation,
    cyclic advection from the grid nodes, sigma.  Returns a dict with ``sigma, x_dep, y_dep, latitude, longitude``."""
    lib = _capi.load()
    dtype = common_dtype(u, v)
    u = np.ascontiguousarray(u, dtype=dtype)
    v = np.ascontiguousarray(v, dtype=dtype)
    if u.shape != v.shape or u.ndim != 3:
        raise ValueError("u and v must both be (time, latitude, longitude)")
    lat_f = np.ascontiguousarray(lat_f, dtype=np.float64)
    lon_f = np.ascontiguousarray(lon_f, dtype=np.float64)
    nt, ny_f, nx_f = u.shape
    if interp_to_common_grid:
        ny, nx = C.c_int(), C.c_int()
        lib.lc_common_grid(C.byref(ny), C.byref(nx), None, None)
        lat, lon = np.empty(ny.value), np.empty(nx.value)
        lib.lc_common_grid(None, None, lat.ctypes.data_as(C.c_void_p), lon.ctypes.data_as(C.c_void_p))
        odt = np.dtype(np.float64)
    else:
        lat, lon, odt = lat_f, lon_f, dtype
    out = {k: np.empty((lat.size, lon.size), odt) for k in ("sigma", "x_dep", "y_dep")}
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    gs = float(gauss_sigma) if isinstance(gauss_sigma, (float, int)) and not isinstance(gauss_sigma, bool) else 0.0
    ctx = C.c_void_p()
    _capi.check(lib.lc_ctx_create(int(device), C.byref(ctx)), lib)
    try:
        if float64_fidelity is not None:
            _capi.check(lib.lc_ctx_set_f64_fidelity(ctx, Engine._FIDELITY[float64_fidelity]), lib)
        _capi.check(lib.lc_lcs_global_host(
            ctx, p(u), p(v), _NP2LC[dtype], nt, ny_f, nx_f, p(lat_f), p(lon_f), int(bool(interp_to_common_grid)),
            -1 if truncation is None else int(truncation), float(timestep), int(SETTLS_order), int(interp_order), gs,
            int(bool(fd_fp32_cast)), _LAYOUTS[tensor_layout], p(out["sigma"]), p(out["x_dep"]), p(out["y_dep"])), lib)
    finally:
        lib.lc_ctx_destroy(ctx)
    out["latitude"], out["longitude"] = lat, lon
    return out
