"""Persistent-workspace driver of the rasterizer C ABI for the training loop and the benchmark.

The reference allocates outputs and the three scratch byte tensors afresh on every call
(src/rasterize_points.cu:68-78,149-157).  On a 288 GB part the trainer keeps them resident instead:
grow-only scratch handed out by the allocator callbacks, gradients written straight into ONE flat
buffer (the RCCL all-reduce bucket / fused-Adam operand), no per-iteration allocation or zero-fill.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _capi

# layout of the flat gradient / parameter bucket, floats per Gaussian (order = FIELDS)
FIELDS = (("means3D", 3), ("scales", 3), ("rotations", 4), ("opacity", 1), ("colors", 3))
FLOATS_PER_GAUSSIAN = sum(n for _, n in FIELDS)


class _GrowBuffer:
    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)

    def callback(self):
        """A fresh ctypes callback per call, not kept on the object (a stored one is a reference cycle: the buffers of a dropped
        engine would wait for the cyclic garbage collector; see rasterize_points._ResizableBuffer)."""
        def _alloc(_ctx, nbytes, box=self):
            if box.tensor.numel() < nbytes:
                box.tensor = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=box.device)
            return box.tensor.data_ptr()
        return _capi.ALLOC_FN(_alloc)


def split_flat(flat: torch.Tensor, P: int):
    """Views (P,n) into a flat FLOATS_PER_GAUSSIAN*P buffer, field-major (each field contiguous)."""
    out, off = {}, 0
    for name, n in FIELDS:
        out[name] = flat[off:off + P * n].view(P, n)
        off += P * n
    return out


class RasterEngine:
    """forward()/backward() over resident buffers; one instance per (P, W, H) on one device.

    resident=True uses the no-host-sync entry points (segs_rasterize_*_resident): the first forward goes through the
    synchronising reference-shaped call to learn R, later calls run with capacity = 1.3 R + slack and only read the
    status words back asynchronously; `check()` (called at the start of the next forward) raises the capacity and
    reports an overflow if R ever outgrew it."""

    def __init__(self, P: int, width: int, height: int, device="cuda:0", resident: bool = False,
                 skip_nonpositive_opacity: bool = False, keep_dead_instances: bool = False, want_cov3D_grad: bool = False):
        self.resident = bool(resident)
        # SEGS_RASTER_SKIP_NONPOSITIVE_OPACITY (segs_raster.h): candidate-domain inputs of segs_neural_forward
        # SEGS_RASTER_KEEP_DEAD_INSTANCES: resident forwards bin the reference's full bounding squares (R == R_reference)
        self.flags = (1 if skip_nonpositive_opacity else 0) | (2 if keep_dead_instances else 0)
        # SEGS_RASTER_EXTRA_FLAGS: A/B measurements only, and only the bit meant for them (16 = SEGS_RASTER_UNFUSED_BINNING,
        # same results); anything else -- a test-support bit, a malformed value -- is ignored rather than changing every engine
        try:
            self.flags |= int(os.environ.get("SEGS_RASTER_EXTRA_FLAGS", "0"), 0) & 16
        except ValueError:
            pass
        self.R_reference = 0
        self.R_live = 0
        self.capacity = 0
        self._status_host = None
        self.P, self.W, self.H = int(P), int(width), int(height)   # P = allocated rows; P_active <= P are rasterized
        self.P_active = self.P
        self.device = torch.device(device)
        f = dict(dtype=torch.float32, device=self.device)
        self.out_color = torch.zeros((3, self.H, self.W), **f)
        self.radii = torch.zeros((self.P,), dtype=torch.int32, device=self.device)
        # (+4 floats behind the bucket: keyframe_parallel.BucketExchange lets the overflow word ride there in a dense exchange)
        self.grads_flat = torch.zeros((FLOATS_PER_GAUSSIAN * self.P + 4,), **f)[:FLOATS_PER_GAUSSIAN * self.P]
        self.grads = split_flat(self.grads_flat, self.P)
        self.dL_dmean2D = torch.zeros((self.P, 3), **f)
        # dL/dcov3D has no consumer when the Gaussians come as scales + rotations (the training path): written on request only;
        # dL/dconic, the tile backward's internal product, is never materialised here (40 B per Gaussian less to write)
        self.dL_dcov3D = torch.zeros((self.P, 6), **f) if want_cov3D_grad else None
        self.geom, self.binning, self.img = (_GrowBuffer(self.device) for _ in range(3))
        self.R = 0
        self._lib = _capi.lib()
        self._last = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_active(self, P_active: int):
        """Rasterize only the first P_active rows of the (P-row) inputs: a map that grows inside pre-sized buffers."""
        assert 0 < P_active <= self.P
        self.P_active = int(P_active)

    # ---- resident mode plumbing
    def _setup_resident(self, R: int):
        # R is the reference-shaped count of the calibrating forward.  Resident forwards bin tight rectangles (typically
        # 0.6-0.75 of it, never more), so 1.25 R leaves them the headroom 1.5 R gives the full lists; the R-sized kernels
        # are launched over the capacity, surplus workgroups cost about 2 us per launch.
        self.capacity = int(R * (1.5 if self.flags & 2 else 1.25)) + 65536
        dev = self.device
        # zero-filled: the resident backward keeps the accumulator rows inside clean itself instead of a fill per iteration
        self._geom_r = torch.zeros(self._lib.segs_geometry_bytes(self.P), dtype=torch.uint8, device=dev)
        self._img_r = torch.empty(self._lib.segs_image_bytes(self.W, self.H), dtype=torch.uint8, device=dev)
        self._bin_r = torch.empty(self._lib.segs_resident_binning_bytes(self.P, self.capacity), dtype=torch.uint8, device=dev)
        self._status = torch.zeros(4, dtype=torch.int32, device=dev)
        self._status_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        self._status_event = torch.cuda.Event()
        self._status_pending = False

    def check(self, raise_on_overflow: bool = True) -> bool:
        """Resolve the last asynchronous status read-back (resident mode).  Returns False (or raises) if the instance
        count outgrew the capacity: that call's outputs are invalid and the next forward re-calibrates."""
        if self.resident and self._status_host is not None and self._status_pending:
            self._status_event.synchronize()
            self._status_pending = False
            self.R = int(self._status_host[0])
            self.R_live = int(self._status_host[1])    # instances the tile kernels walk (dead ones dropped by the first sort pass)
            if int(self._status_host[3]) != 0:
                need = self.R
                self.capacity = 0  # next forward goes through the synchronising path and sizes the scratch anew
                if raise_on_overflow:
                    raise RuntimeError(f"resident rasterizer: {need} instances exceeded the capacity; outputs of that step are invalid")
                return False
        return True

    def poll(self) -> bool:
        """check() without waiting: resolves the last status read-back only if the device has got there.  For the hipGraph
        replay loop, where waiting for the previous step's event would serialise host and device."""
        if self.resident and self._status_host is not None and self._status_pending and self._status_event.query():
            return self.check(raise_on_overflow=False)
        return True

    def after_graph_replay(self):
        """Bookkeeping of a resident forward + backward that ran from a captured graph (the Python side of forward() did not)."""
        self._status_event.record(torch.cuda.current_stream(self.device))
        self._status_pending = True
        self._last_resident = True

    def forward(self, bg, means3D, colors, opacity, scales, rotations, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                scale_modifier: float = 1.0) -> torch.Tensor:
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        for t in (bg, means3D, colors, opacity, scales, rotations, viewmatrix, projmatrix, campos):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
        if self.resident and self.capacity > 0:
            self.check(raise_on_overflow=False)  # an overflow noticed here was already handled by the caller's own check
        # the two switches are per-host-thread state of the library: set for this call only, restored afterwards, so that
        # the reference-shaped wrappers (rasterize_points.py) keep the reference's behaviour on the same thread
        old_flags = self._lib.segs_raster_set_flags(self.flags)
        try:
            return self._forward(p, bg, means3D, colors, opacity, scales, rotations, viewmatrix, projmatrix, campos, tanfovx,
                                 tanfovy, scale_modifier)
        finally:
            self._lib.segs_raster_set_flags(old_flags)
            self._lib.segs_raster_set_status_mirror(None)

    def _forward(self, p, bg, means3D, colors, opacity, scales, rotations, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                 scale_modifier):
        if self.resident and self.capacity > 0:
            self._lib.segs_raster_set_status_mirror(C.c_void_p(self._status_host.data_ptr()))
            st = self._lib.segs_rasterize_forward_resident(
                p(self._geom_r), p(self._bin_r), p(self._img_r), self.capacity, self.P, self.P_active, 0, 0, p(bg), self.W, self.H, p(means3D),
                None, p(colors), p(opacity), p(scales), float(scale_modifier), p(rotations), None, p(viewmatrix), p(projmatrix),
                p(campos), float(tanfovx), float(tanfovy), p(self.out_color), p(self.radii), p(self._status), self._stream())
            _capi.check(st, "segs_rasterize_forward_resident")
            if not torch.cuda.is_current_stream_capturing():
                # R and the overflow word were stored into the pinned host words by the last binning kernel
                self._status_event.record(torch.cuda.current_stream(self.device))
                self._status_pending = True
            self._last = (bg, means3D, colors, opacity, scales, rotations, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                          scale_modifier)
            self._last_resident = True
            return self.out_color
        self._last_resident = False
        n = C.c_int(0)
        gcb, bcb, icb = self.geom.callback(), self.binning.callback(), self.img.callback()
        st = self._lib.segs_rasterize_forward(
            gcb, None, bcb, None, icb, None, self.P_active, 0, 0, p(bg), self.W, self.H, p(means3D),
            None, p(colors), p(opacity), p(scales), float(scale_modifier), p(rotations), None, p(viewmatrix), p(projmatrix),
            p(campos), float(tanfovx), float(tanfovy), 0, p(self.out_color), p(self.radii), self._stream(), C.byref(n))
        _capi.check(st, "segs_rasterize_forward")
        self.R = int(n.value)
        self.R_reference = self.R     # the reference's num_rendered (bounding-square duplication, rasterizer_impl.cu:70-111)
        self._last = (bg, means3D, colors, opacity, scales, rotations, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                      scale_modifier)
        if self.resident and self.capacity == 0:
            self._setup_resident(self.R)  # calibrated: later forwards take the no-sync path
        return self.out_color

    def can_take_projected(self) -> bool:
        """The resident buffers are calibrated: a producer may run K1 itself (projection_targets / forward_projected)."""
        return bool(self.resident and self.capacity > 0)

    def projection_targets(self) -> "_capi.ProjectionTargets":
        """Where a producer that projects its own Gaussians leaves K1's outputs for the next forward_projected call
        (segs_resident_projection_targets); made under this engine's flags."""
        assert self.can_take_projected()
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        tg = _capi.ProjectionTargets()
        old_flags = self._lib.segs_raster_set_flags(self.flags)
        try:
            st = self._lib.segs_resident_projection_targets(p(self._geom_r), p(self._bin_r), p(self._img_r), self.capacity, self.P,
                                                            self.P_active, self.W, self.H, p(self.radii), p(self._status), C.byref(tg))
        finally:
            self._lib.segs_raster_set_flags(old_flags)
        _capi.check(st, "segs_resident_projection_targets")
        return tg

    def forward_projected(self, bg, means3D, scales, rotations, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                          scale_modifier: float = 1.0) -> torch.Tensor:
        """The resident forward WITHOUT its per-Gaussian stage: the producer has written records, radii, tile counts and depth
        keys into projection_targets().  means3D / scales / rotations are what the backward will re-read."""
        assert self.can_take_projected()   # (the caller resolved the previous step's status before it asked for the targets)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        old_flags = self._lib.segs_raster_set_flags(self.flags)
        try:
            self._lib.segs_raster_set_status_mirror(C.c_void_p(self._status_host.data_ptr()))
            st = self._lib.segs_rasterize_forward_resident_projected(p(self._geom_r), p(self._bin_r), p(self._img_r), self.capacity, self.P,
                                                                     self.P_active, p(bg), self.W, self.H, p(self.out_color),
                                                                     p(self._status), self._stream())
            _capi.check(st, "segs_rasterize_forward_resident_projected")
        finally:
            self._lib.segs_raster_set_flags(old_flags)
            self._lib.segs_raster_set_status_mirror(None)
        if not torch.cuda.is_current_stream_capturing():
            self._status_event.record(torch.cuda.current_stream(self.device))
            self._status_pending = True
        self._last = (bg, means3D, None, None, scales, rotations, viewmatrix, projmatrix, campos, tanfovx, tanfovy, scale_modifier)
        self._last_resident = True
        return self.out_color

    def backward(self, dL_dout_color: torch.Tensor):
        """Gradients land in self.grads (views of self.grads_flat), dL_dmean2D, dL_dcov3D."""
        (bg, means3D, colors, opacity, scales, rotations, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
         scale_modifier) = self._last
        assert dL_dout_color.is_contiguous() and dL_dout_color.dtype == torch.float32
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        pn = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        g = self.grads
        if getattr(self, "_last_resident", False):
            st = self._lib.segs_rasterize_backward_resident(
                p(self._geom_r), p(self._bin_r), p(self._img_r), self.capacity, self.P, self.P_active, 0, 0, p(bg), self.W, self.H, p(means3D),
                None, p(scales), float(scale_modifier), p(rotations), None, p(viewmatrix), p(projmatrix), p(campos),
                float(tanfovx), float(tanfovy), p(self.radii), p(dL_dout_color), p(self.dL_dmean2D), None,
                p(g["opacity"]), p(g["colors"]), p(g["means3D"]), pn(self.dL_dcov3D), None, p(g["scales"]), p(g["rotations"]),
                self._stream())
            _capi.check(st, "segs_rasterize_backward_resident")
            return g
        st = self._lib.segs_rasterize_backward(
            self.P_active, 0, 0, self.R, p(bg), self.W, self.H, p(means3D), None, p(colors), p(scales), float(scale_modifier),
            p(rotations), None, p(viewmatrix), p(projmatrix), p(campos), float(tanfovx), float(tanfovy), p(self.radii),
            p(self.geom.tensor), p(self.binning.tensor), p(self.img.tensor), p(dL_dout_color), p(self.dL_dmean2D),
            None, p(g["opacity"]), p(g["colors"]), p(g["means3D"]), pn(self.dL_dcov3D), None, p(g["scales"]),
            p(g["rotations"]), self._stream())
        _capi.check(st, "segs_rasterize_backward")
        return g


class KernelProfile:
    """Context manager around segs_profile_begin/end: per-kernel HIP-event times of the calls inside."""

    def __init__(self, names=None):
        self._lib = _capi.lib()
        n = self._lib.segs_profile_kernel_count()
        self.names = [self._lib.segs_profile_kernel_name(i).decode() for i in range(n)]
        self.mask = sum(1 << i for i, nm in enumerate(self.names) if names is None or nm in names)
        self.result = {}

    def __enter__(self):
        _capi.check(self._lib.segs_profile_begin(self.mask), "segs_profile_begin")
        return self

    def __exit__(self, *exc):
        _capi.check(self._lib.segs_profile_end(), "segs_profile_end")
        for i, nm in enumerate(self.names):
            ms, cnt = C.c_double(0), C.c_long(0)
            self._lib.segs_profile_query(i, C.byref(ms), C.byref(cnt))
            if cnt.value:
                self.result[nm] = dict(total_ms=ms.value, launches=cnt.value, avg_ms=ms.value / cnt.value)
        return False
