"""ctypes binding of the C ABI in include/arah_hip.h (libarah_hip.so, built by __graft_entry__.build()).

Only raw device pointers, sizes and the current HIP stream cross the boundary; torch is used for
device memory and streams.  There is NO fallback: if the shared library is missing or no GPU is
present, every entry point raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARAH_LIB_PATH: load another build of the same ABI (the instrumented one of tools/phase_clocks.py)
LIB_PATH = os.environ.get("ARAH_LIB_PATH") or os.path.join(_HERE, "libarah_hip.so")

ARAH_MAX_STEPS = 128
COLOR_NO_VIEW_DIR = 0
COLOR_IDR = 1
PRECISION_SPLIT_F16 = 0   # fp32 carried as hi + lo f16 pairs on the f16 matrix pipe (default)
PRECISION_FP32 = 1        # v_mfma_f32_16x16x4_f32 everywhere


SHADE_ENGINE_DEFAULT, SHADE_ENGINE_FP32 = 0, 1
CANON_KERNEL_WAVE, CANON_KERNEL_TILE, CANON_KERNEL_WAVE_L2 = 0, 1, 2


def default_shade_engine():
    """ARAH_SHADE_ENGINE=fp32: loop D's normal sweep and colour MLP on the fp32 MFMA; default bf16 x 3 (split frames)."""
    return SHADE_ENGINE_FP32 if os.environ.get("ARAH_SHADE_ENGINE", "b3") == "fp32" else SHADE_ENGINE_DEFAULT


def default_canon_kernel():
    """ARAH_CANON_KERNEL = wave (default) | tile | wave_l2: loop C's solver on split-engine frames."""
    return {"tile": CANON_KERNEL_TILE, "wave_l2": CANON_KERNEL_WAVE_L2}.get(os.environ.get("ARAH_CANON_KERNEL", "wave"), CANON_KERNEL_WAVE)


def default_precision():
    """ARAH_PRECISION=fp32 selects the exact engine for every GEMM; default is the split engine."""
    v = os.environ.get("ARAH_PRECISION", "split").lower()
    if v in ("split", "split_f16", "f16x3", "0"):
        return PRECISION_SPLIT_F16
    if v in ("fp32", "f32", "exact", "1"):
        return PRECISION_FP32
    raise ValueError("ARAH_PRECISION must be 'split' or 'fp32', got %r" % v)

_ERRORS = {-1: "ARAH_E_BADARG", -2: "ARAH_E_SHAPE", -3: "ARAH_E_WORKSPACE", -4: "ARAH_E_LAUNCH",
           -5: "ARAH_E_SAMPLING"}

_fp = C.c_void_p  # device float*


class ArahNets(C.Structure):
    _fields_ = [("sdf_w", _fp * 7), ("sdf_b", _fp * 7), ("film_freq", _fp), ("film_phase", _fp),
                ("skin_w", _fp * 5), ("skin_b", _fp * 5), ("col_w", _fp * 6), ("col_b", _fp * 6),
                ("pose_vec", _fp), ("col_mode", C.c_int32), ("n_pose", C.c_int32), ("beta", _fp),
                ("precision", C.c_int32)]


class ArahBody(C.Structure):
    _fields_ = [("verts", _fp), ("vert_weights", _fp), ("bones", _fp), ("trans", _fp), ("center", _fp),
                ("coord_min", _fp), ("coord_max", _fp), ("n_verts", C.c_int32), ("prepared", C.c_void_p)]


class ArahSampling(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("n_near", C.c_int32), ("n_far", C.c_int32),
                ("cano_view_dirs", C.c_int32), ("render_last_pt", C.c_int32), ("full_shading", C.c_int32),
                ("lin_steps", _fp), ("lin_near", _fp), ("lin_far", _fp),
                ("shade_engine", C.c_int32), ("canon_kernel", C.c_int32),
                ("ev_canon", C.c_void_p * 2), ("ev_density", C.c_void_p * 2), ("ev_shade", C.c_void_p * 2),
                ("occupancy", C.c_void_p), ("ev_canon2", C.c_void_p * 2), ("ev_density2", C.c_void_p * 2)]


class ArahFrame(C.Structure):
    _fields_ = [("sdf_w0", _fp), ("sdf_wp", _fp * 5), ("sdf_wpT", _fp * 5), ("sdf_w6", _fp), ("sdf_b6", _fp),
                ("sdf_bias", _fp), ("sdf_freq", _fp), ("sdf_phase", _fp),
                ("sdf_wps", _fp * 5), ("sdf_fw", _fp), ("sdf_pw", _fp), ("sdf_fws", _fp),
                ("skin_w0", _fp), ("skin_wp", _fp * 3), ("skin_w4p", _fp), ("skin_bias", _fp),
                ("skin_wps", _fp * 4), ("skin_scales", _fp), ("skin_wpr", _fp), ("skin_wconsts", _fp),
                ("col_w0p", _fp), ("col_w1p", _fp), ("col_w2p", _fp), ("col_w3ap", _fp), ("col_w3bp", _fp),
                ("col_w4p", _fp), ("col_w5", _fp), ("col_bias", _fp),
                ("col_w0pT", _fp), ("col_w1pT", _fp), ("col_w2pT", _fp), ("col_w3apT", _fp), ("col_w3bpT", _fp),
                ("col_w4pT", _fp), ("b3", C.c_void_p * 22),
                ("verts4", _fp), ("knn_spheres", _fp), ("knn_grid", _fp), ("knn_cells", _fp),
                ("verts", _fp), ("vert_T", _fp), ("bones", _fp),
                ("scalars", _fp), ("n_verts", C.c_int32),
                ("col_mode", C.c_int32), ("precision", C.c_int32)]


class ArahTrainIn(C.Structure):
    _fields_ = [("n", C.c_int32), ("rotate_normal", C.c_int32), ("ray_augm", C.c_int32), ("geom_only", C.c_int32),
                ("x", _fp), ("T", _fp), ("view", _fp), ("view_orig", _fp), ("g_s", _fp), ("g_rgb", _fp),
                ("tap_cin", _fp), ("tap_c", _fp * 5), ("fwd_rgb4", _fp)]


class ArahTrainGrads(C.Structure):
    _fields_ = [("sdf", _fp), ("rgb4", _fp), ("gx4", _fp), ("film_freq", _fp), ("film_phase", _fp),
                ("h", _fp * 6), ("hd", _fp * 7), ("av", _fp * 6), ("avd", _fp * 6), ("cin", _fp), ("c", _fp * 5),
                ("d", _fp * 6)]


class ArahCounters(C.Structure):
    _fields_ = [("n_sdf_fwd", C.c_uint64), ("n_sdf_grad", C.c_uint64), ("n_skin_fwd", C.c_uint64),
                ("n_skin_jac", C.c_uint64), ("n_col", C.c_uint64), ("n_knn", C.c_uint64),
                ("n_density", C.c_uint64), ("n_canon", C.c_uint64), ("n_split_nonfinite", C.c_uint64),
                ("n_tier_rays", C.c_uint64), ("n_tier_rays_surface", C.c_uint64), ("n_tier_rays_promoted", C.c_uint64),
                ("n_tier_rays_skipped", C.c_uint64), ("n_tier_samples_p1", C.c_uint64), ("n_tier_samples_p2", C.c_uint64),
                ("n_tier_samples_skipped", C.c_uint64), ("n_tier_witnesses", C.c_uint64),
                ("n_tier_rays_untraced", C.c_uint64),
                ("n_canon_p2", C.c_uint64), ("n_density_p2", C.c_uint64)]


COUNTER_BYTES = C.sizeof(ArahCounters)

EXPORTS = ["arah_frame_bytes", "arah_prepare_frame", "arah_body_bytes", "arah_prepare_body", "arah_workspace_bytes", "arah_counters_reset",
           "arah_counters_read", "arah_sdf_eval", "arah_sdf_grid", "arah_rasterize", "arah_skin_lbs", "arah_skin_jacobian", "arah_color_eval",
           "arah_nearest_inverse_lbs", "arah_broyden3_lbs", "arah_joint_root_find", "arah_trace", "arah_sample_canonicalize",
           "arah_shade_composite", "arah_shade_points", "arah_render", "arah_shade_train_slab_bytes", "arah_shade_train_forward",
           "arah_shade_train_backward", "arah_composite_train_forward", "arah_composite_train_backward", "arah_gram_skinny_blocks", "arah_gram_skinny", "arah_colsum_blocks", "arah_colsum", "arah_inverse3x3", "arah_hsoftmax_train_forward", "arah_hsoftmax_train_backward", "arah_pose_tree_forward", "arah_pose_tree_backward", "arah_gemv_rows", "arah_mesh_query_scratch_bytes", "arah_mesh_query", "arah_dominant_kernel",
           "arah_skin_lbs_counted", "arah_marching_cubes_scratch_bytes", "arah_marching_cubes",
           "arah_occupancy_bytes", "arah_prepare_occupancy", "arah_occupancy_info", "arah_tier_debug", "arah_debug_samples",
           "arah_sdf_grid_band_scratch_bytes", "arah_sdf_grid_band"]

_lib = None


def load_library():
    """dlopen libarah_hip.so (after torch, so that it binds to the HIP runtime torch already loaded)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libarah_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` -- there is no CPU fallback for the hot path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.arah_frame_bytes.restype = C.c_size_t
    lib.arah_body_bytes.restype = C.c_size_t
    lib.arah_workspace_bytes.restype = C.c_size_t
    lib.arah_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.arah_dominant_kernel.restype = C.c_char_p
    lib.arah_shade_train_slab_bytes.restype = C.c_size_t
    lib.arah_mesh_query_scratch_bytes.restype = C.c_size_t
    lib.arah_marching_cubes_scratch_bytes.restype = C.c_size_t
    lib.arah_marching_cubes_scratch_bytes.argtypes = [C.c_int32]
    lib.arah_occupancy_bytes.restype = C.c_size_t
    lib.arah_sdf_grid_band_scratch_bytes.restype = C.c_size_t
    lib.arah_colsum_blocks.argtypes = [C.c_int64]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if the symbol is missing
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, _ERRORS.get(rc, "?"), rc))


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(dtype=torch.float32).contiguous()


_call_device = [None]   # device of the C-ABI call in flight (set by _guarded / Frame.__init__)


def _stream(device=None):
    """Current HIP stream OF the call's device (not of whatever device happens to be current)."""
    device = device if device is not None else _call_device[0]
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _on_device:
    """Make `device` current for the duration of a C-ABI call: the kernels, memsets and copies behind the ABI are
    issued on the current device, torch's own guard does not reach through ctypes."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.guard = torch.cuda.device(self.device)

    def __enter__(self):
        self.guard.__enter__()
        self.prev = _call_device[0]
        _call_device[0] = self.device

    def __exit__(self, *exc):
        _call_device[0] = self.prev
        return self.guard.__exit__(*exc)


def _guarded(fn):
    """Wrapper for fn(frame, ws, ...): frame, workspace and every tensor argument must share one GPU, which
    becomes the current device (and supplies the stream) for the call."""
    import functools

    @functools.wraps(fn)
    def wrapper(frame, ws, *args, **kwargs):
        dev = frame.device
        if ws.device != dev:
            raise ValueError("workspace lives on %s, frame on %s" % (ws.device, dev))

        def check(v):
            if isinstance(v, torch.Tensor):
                if v.device != dev:
                    raise ValueError("argument on %s, frame on %s" % (v.device, dev))
            elif isinstance(v, (tuple, list)):
                for u in v:
                    check(u)
        for v in list(args) + list(kwargs.values()):
            check(v)
        with _on_device(dev):
            return fn(frame, ws, *args, **kwargs)
    return wrapper


def _same_device(*tensors):
    """All device buffers of one C-ABI call must live on one GPU; returns it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise ValueError("device-resident tensor required")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError("buffers of one call live on different devices: %s vs %s" % (dev, t.device))
    return dev


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("arah_release_amd needs a HIP device (MI355X / gfx950); none is visible and there "
                           "is no CPU fallback")


class Workspace:
    """Caller-owned scratch of the C ABI (grows on demand, never shrinks)."""

    def __init__(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.buf = None
        self.occ = None   # occupancy bitmap of the frame this scratch is rendering (tiered eval forward)

    def occupancy(self, frame):
        """arah_prepare_occupancy for `frame` on the current stream, into this scratch's own buffer (a scratch serves one
        stream: the bitmap of the previous frame is dead by stream order when the next one is built)."""
        lib = load_library()
        buf = self.ensure(1, 1)
        with torch.cuda.device(self.device):
            if self.occ is None:
                self.occ = torch.empty(int(lib.arah_occupancy_bytes()), dtype=torch.uint8, device=self.device)
            _check(lib.arah_prepare_occupancy(C.byref(frame.handle), _ptr(self.occ), C.c_size_t(self.occ.numel()), _ptr(buf),
                                              C.c_size_t(buf.numel()), _stream(self.device)), "arah_prepare_occupancy")
        return self.occ

    def occupancy_info(self):
        """Header of the last occupancy built here (synchronises the stream): dict of its geometry and counts."""
        import struct
        out = (C.c_int32 * 16)()
        with torch.cuda.device(self.device):
            _check(load_library().arah_occupancy_info(_ptr(self.occ), out, _stream(self.device)), "arah_occupancy_info")
        raw = bytes(out)
        f = struct.unpack("5f", raw[:20])
        i = struct.unpack("9i", raw[20:56])
        return {"origin": f[:3], "voxel": f[3], "dims": i[:3], "n_vox": i[3], "valid": i[4], "n_cells": i[5], "n_fine": i[6],
                "n_selected": i[7], "overflow": i[8], "band_m": struct.unpack("f", raw[56:60])[0],
                "lip_pose": struct.unpack("f", raw[60:64])[0]}

    def tier_debug(self, n_rays, n_steps):
        """(ray_tier, ray_sigma_pos) uint8 tensors of the last arah_render on this scratch."""
        tier = torch.empty(n_rays, dtype=torch.uint8, device=self.device)
        pos = torch.empty(n_rays, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _check(load_library().arah_tier_debug(_ptr(self.buf), C.c_size_t(self.buf.numel()), C.c_int32(n_rays), C.c_int32(n_steps),
                                                  _ptr(tier), _ptr(pos), _stream(self.device)), "arah_tier_debug")
        return tier, pos

    def debug_samples(self, n_rays, n_steps, which=("z", "pts", "T", "mask", "shaded", "state")):
        """Per-sample arrays of the last arah_render on this scratch: dict of the tensors named in `which`."""
        d, Q = self.device, n_rays * n_steps
        make = {"z": lambda: torch.empty(Q, device=d), "pts": lambda: torch.empty(Q, 3, device=d), "T": lambda: torch.empty(Q, 16, device=d),
                "mask": lambda: torch.empty(Q, dtype=torch.uint8, device=d), "shaded": lambda: torch.empty(Q, 4, device=d),
                "state": lambda: torch.empty(Q, dtype=torch.uint8, device=d)}
        out = {k: make[k]() for k in which}
        with torch.cuda.device(d):
            _check(load_library().arah_debug_samples(_ptr(self.buf), C.c_size_t(self.buf.numel()), C.c_int32(n_rays), C.c_int32(n_steps),
                                                     _ptr(out.get("z")), _ptr(out.get("pts")), _ptr(out.get("T")), _ptr(out.get("mask")),
                                                     _ptr(out.get("shaded")), _ptr(out.get("state")), _stream(d)), "arah_debug_samples")
        return out

    def ensure(self, n_rays, n_steps):
        need = load_library().arah_workspace_bytes(int(n_rays), int(n_steps))
        if self.buf is None or self.buf.numel() < need:
            old = self.buf
            with torch.cuda.device(self.device):
                self.buf = torch.empty(need, dtype=torch.uint8, device=self.device)
                if old is None:
                    _check(load_library().arah_counters_reset(_ptr(self.buf), _stream(self.device)),
                           "arah_counters_reset")
                else:   # the work counters sit at the head of the workspace: a grown buffer inherits them
                    self.buf[:COUNTER_BYTES].copy_(old[:COUNTER_BYTES])
        return self.buf

    def reset_counters(self):
        with torch.cuda.device(self.device):
            _check(load_library().arah_counters_reset(_ptr(self.buf), _stream(self.device)), "arah_counters_reset")

    def counters(self):
        out = ArahCounters()
        with torch.cuda.device(self.device):
            _check(load_library().arah_counters_read(_ptr(self.buf), C.byref(out), _stream(self.device)),
                   "arah_counters_read")
        return {k: int(getattr(out, k)) for k, _ in ArahCounters._fields_}


_side_streams = {}


class BodyTables:
    """The nearest-vertex tables of one posed body (k-d clustered vertices, cluster spheres, per-cell candidate lists),
    built by arah_prepare_body on a SIDE stream: the 1.4 ms of a one-workgroup sort and a 512-workgroup pass then run next to
    whatever the caller's stream does in the meantime (the pose encoder and the hypernetwork: the posed vertices are
    known before them).  Frame(..., body_tables=...) orders the side stream before its own and keeps this object alive."""

    def __init__(self, verts):
        require_gpu()
        lib = load_library()
        dev = verts.device
        self.verts = _f32(verts)
        self.verts_version = self.verts._version
        if self.verts.dim() != 2 or self.verts.shape[1] != 3:
            raise ValueError("verts must be (V, 3)")
        cur = torch.cuda.current_stream(dev)
        side = _side_streams.get(dev.index)
        if side is None:
            side = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
        with _on_device(dev):
            self.buf = torch.empty(int(lib.arah_body_bytes()), dtype=torch.uint8, device=dev)
            side.wait_stream(cur)                       # the vertices (and the recycled buffer) are the caller's stream's
            self.buf.record_stream(side)
            self.verts.record_stream(side)
            _check(lib.arah_prepare_body(_ptr(self.verts), C.c_int32(int(self.verts.shape[0])), _ptr(self.buf),
                                         C.c_size_t(self.buf.numel()), C.c_void_p(side.cuda_stream)), "arah_prepare_body")
            self.done = torch.cuda.Event()
            self.done.record(side)
        self.device = dev


class Frame:
    """Packed per-frame state on the device (MFMA-ordered weights, padded vertices)."""

    def __init__(self, sdf_layers, film_freq, film_phase, skin_layers, color_layers, color_mode, pose_vec, beta,
                 verts, vert_weights, bones, trans, center, coord_min, coord_max, precision=None, body_tables=None):
        require_gpu()
        lib = load_library()
        dev = verts.device
        keep = []  # tensors referenced by raw pointers must outlive the call

        def own(t):
            t = _f32(t)
            keep.append(t)
            return t

        def scalar_tensor(v, n):
            """Per-frame scalars (trans, center, coord_min/max, |variance|) stay on the device: tensors are used as
            they are (no .item() / .tolist(): those drain the stream); python numbers are uploaded."""
            if isinstance(v, torch.Tensor):
                return v.to(device=dev, dtype=torch.float32).reshape(-1)[:n]
            return torch.tensor(v, dtype=torch.float32, device=dev).reshape(-1)[:n]

        nets = ArahNets()
        if color_layers is None:   # tracer-only use: the colour MLP is never evaluated
            color_mode, pose_vec = COLOR_NO_VIEW_DIR, None
            dims = [(256, 262), (256, 256), (128, 256), (256, 390), (256, 256), (3, 256)]
            color_layers = [(torch.zeros(d, device=dev), torch.zeros(d[0], device=dev)) for d in dims]
        if len(sdf_layers) != 7 or len(skin_layers) != 5 or len(color_layers) != 6:
            raise ValueError("unsupported network depth (ARAH_E_SHAPE)")
        shapes_sdf = [(256, 3)] + [(256, 256)] * 5 + [(1, 256)]
        for i, (w, b) in enumerate(sdf_layers):
            if tuple(w.shape) != shapes_sdf[i]:
                raise ValueError("SDF layer %d has shape %s, expected %s" % (i, tuple(w.shape), shapes_sdf[i]))
            nets.sdf_w[i] = _ptr(own(w)).value
            nets.sdf_b[i] = _ptr(own(b)).value
        nets.film_freq = _ptr(own(film_freq.reshape(-1)))
        nets.film_phase = _ptr(own(film_phase.reshape(-1)))
        shapes_skin = [(128, 3)] + [(128, 128)] * 3 + [(25, 128)]
        for i, (w, b) in enumerate(skin_layers):
            if tuple(w.shape) != shapes_skin[i]:
                raise ValueError("skinning layer %d has shape %s, expected %s" % (i, tuple(w.shape), shapes_skin[i]))
            nets.skin_w[i] = _ptr(own(w)).value
            nets.skin_b[i] = _ptr(own(b)).value
        n_pose = 0 if pose_vec is None else int(pose_vec.numel())
        in_dim = (33 if color_mode == COLOR_IDR else 6) + 256 + n_pose
        shapes_col = [(256, in_dim), (256, 256), (128, 256), (256, in_dim + 128), (256, 256), (3, 256)]
        for i, (w, b) in enumerate(color_layers):
            if tuple(w.shape) != shapes_col[i]:
                raise ValueError("colour layer %d has shape %s, expected %s" % (i, tuple(w.shape), shapes_col[i]))
            nets.col_w[i] = _ptr(own(w)).value
            nets.col_b[i] = _ptr(own(b)).value
        nets.pose_vec = _ptr(own(pose_vec.reshape(-1))) if n_pose else None
        nets.col_mode = int(color_mode)
        nets.n_pose = n_pose
        nets.beta = _ptr(own(scalar_tensor(beta, 1)))
        nets.precision = default_precision() if precision is None else int(precision)
        self.precision = nets.precision
        body = ArahBody()
        self.verts, self.vert_weights, self.bones = own(verts), own(vert_weights), own(bones.reshape(24, 16))
        body.verts, body.vert_weights, body.bones = _ptr(self.verts), _ptr(self.vert_weights), _ptr(self.bones)
        body.trans, body.center = _ptr(own(scalar_tensor(trans, 3))), _ptr(own(scalar_tensor(center, 3)))
        body.coord_min, body.coord_max = _ptr(own(scalar_tensor(coord_min, 1))), _ptr(own(scalar_tensor(coord_max, 1)))
        body.n_verts = int(verts.shape[0])
        self.body_tables = body_tables
        if body_tables is not None:
            # the tables must be THESE vertices': same storage, untouched since (a reused inputs dict whose smpl_verts were
            # replaced or modified in place after the tables were built would otherwise be searched with a stale body)
            same = tuple(body_tables.verts.shape) == tuple(self.verts.shape) and (
                (body_tables.verts.data_ptr() == self.verts.data_ptr() and body_tables.verts_version == self.verts._version)
                or bool(torch.equal(body_tables.verts, self.verts)))
            if body_tables.device != dev or not same:
                raise ValueError("body_tables were built for another body / device")
            torch.cuda.current_stream(dev).wait_event(body_tables.done)
            body.prepared = body_tables.buf.data_ptr()
        _same_device(*keep)
        nbytes = lib.arah_frame_bytes(C.byref(nets), C.byref(body))
        self.handle = ArahFrame()
        with _on_device(dev):
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _check(lib.arah_prepare_frame(C.byref(nets), C.byref(body), _ptr(self.buf), C.c_size_t(nbytes),
                                          C.byref(self.handle), _stream()), "arah_prepare_frame")
        # the pack kernels read `keep` asynchronously on the current stream; hold the raw tensors
        # until the frame is dropped (cheap: a few MB)
        self._keep = keep
        self.device = dev
        self.color_mode = int(color_mode)


class Sampling:
    """ArahSampling + the device linspace tables (bit-identical to torch.linspace on the CPU)."""

    def __init__(self, device, n_steps=64, n_near=16, n_far=16, cano_view_dirs=True, render_last_pt=False,
                 full_shading=False, shade_engine=None, canon_kernel=None):
        if n_steps < n_near + n_far + 1 or n_steps > ARAH_MAX_STEPS:
            raise ValueError("need n_near + n_far + 1 <= n_steps <= %d (ARAH_E_SAMPLING)" % ARAH_MAX_STEPS)
        self.lin_steps = torch.linspace(0.0, 1.0, n_steps, dtype=torch.float32).to(device)
        self.lin_near = torch.linspace(0.0, 1.0, n_near + 1, dtype=torch.float32).to(device)
        self.lin_far = torch.linspace(0.0, 1.0, max(n_far, 1), dtype=torch.float32).to(device)
        s = ArahSampling()
        s.n_steps, s.n_near, s.n_far = int(n_steps), int(n_near), int(n_far)
        s.cano_view_dirs, s.render_last_pt = int(bool(cano_view_dirs)), int(bool(render_last_pt))
        s.full_shading = int(bool(full_shading))
        s.lin_steps, s.lin_near, s.lin_far = _ptr(self.lin_steps), _ptr(self.lin_near), _ptr(self.lin_far)
        # per-call switches of the library (it reads no environment itself): None = this process's default
        s.shade_engine = default_shade_engine() if shade_engine is None else int(shade_engine)
        s.canon_kernel = default_canon_kernel() if canon_kernel is None else int(canon_kernel)
        self.handle = s
        self._events = {}
        self.n_steps, self.n_near, self.n_far = n_steps, n_near, n_far

    def set_events(self, which, start=None, stop=None):
        """Profiling hook of THIS sampling object: every call that takes it records the torch.cuda.Event pair on its stream
        around loop C's solver ("canon"), the density pre-pass ("density") or the shading kernel ("shade").  None switches
        it off.  (Round 3 kept the handles in process-wide variables of the library.)"""
        field = getattr(self.handle, {"canon": "ev_canon", "density": "ev_density", "shade": "ev_shade", "canon2": "ev_canon2",
                                      "density2": "ev_density2"}[which])
        for ev in (start, stop):   # torch creates the hipEvent lazily, on the first record
            if ev is not None and not ev.cuda_event:
                ev.record()
        field[0] = start.cuda_event if start is not None else None
        field[1] = stop.cuda_event if stop is not None else None
        self._events[which] = (start, stop)   # keep them alive


# ------------------------------------------------------------------------------------------------
# thin functional wrappers (allocate outputs with torch, call the C ABI on the current stream)
# ------------------------------------------------------------------------------------------------
@_guarded
def sdf_eval(frame, ws, x_norm, want_feat=False, want_grad=False):
    lib = load_library()
    x = _f32(x_norm)
    n = x.shape[0]
    buf = ws.ensure(1, 1)
    sdf = torch.empty(n, device=x.device)
    feat = torch.empty(n, 256, device=x.device) if want_feat else None
    grad = torch.empty(n, 3, device=x.device) if want_grad else None
    _check(lib.arah_sdf_eval(C.byref(frame.handle), _ptr(x), C.c_int32(n), _ptr(sdf), _ptr(feat), _ptr(grad),
                             _ptr(buf), C.c_size_t(buf.numel()), _stream()), "arah_sdf_eval")
    return sdf, feat, grad


@_guarded
def sdf_grid(frame, ws, n_side=256):
    """SDF on the n_side^3 lattice of [-1,1]^3 -> (n_side, n_side, n_side) tensor [ix, iy, iz], on the device."""
    lib = load_library()
    buf = ws.ensure(1, 1)
    out = torch.empty(n_side, n_side, n_side, device=frame.device)
    _check(lib.arah_sdf_grid(C.byref(frame.handle), C.c_int32(int(n_side)), _ptr(out), _ptr(buf), C.c_size_t(buf.numel()),
                             _stream()), "arah_sdf_grid")
    return out


_band_scratch = {}


@_guarded
def sdf_grid_band(frame, ws, n_side=256):
    """The lattice of sdf_grid for marching cubes at level 0: exact where the level set can pass, the right sign elsewhere
    (csrc/tier.hpp; ~6 % of the evaluations).  -> (volume (n_side,)*3, n_evaluated (1,) int32 on the device)."""
    lib = load_library()
    buf = ws.ensure(1, 1)
    dev = frame.device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, int(n_side))
    sc = _band_scratch.get(key)
    if sc is None:
        if len(_band_scratch) >= 8:
            _band_scratch.pop(next(iter(_band_scratch)))
        sc = _band_scratch[key] = (torch.empty(int(lib.arah_sdf_grid_band_scratch_bytes()), dtype=torch.uint8, device=dev),
                                   torch.empty(n_side ** 3, dtype=torch.int32, device=dev))
    scratch, lst = sc
    out = torch.empty(n_side, n_side, n_side, device=dev)
    _check(lib.arah_sdf_grid_band(C.byref(frame.handle), C.c_int32(int(n_side)), _ptr(out), _ptr(lst), _ptr(scratch),
                                  C.c_size_t(scratch.numel()), _ptr(buf), C.c_size_t(buf.numel()), _stream()), "arah_sdf_grid_band")
    n_eval = scratch[0:4].view(torch.int32)
    return out, n_eval


_mc_tables = {}


def marching_cubes(sdf, level=0.0, cap=1 << 20):
    """Level set of the lattice volume sdf (N,N,N) [ix,iy,iz] as a triangle soup, on the device and WITHOUT a host round
    trip: -> tris (cap,3,3) coordinates in [-1,1]^3 (rows beyond the count are zero: degenerate triangles), n_tris (1,) int32
    on the device (the size of the level set; it may exceed cap, then only the first cap triangles were written).  The case
    table is meshing.case_table(), copied to the device once; the result is meshing.marching_cubes(sdf) triangle for triangle."""
    require_gpu()
    lib = load_library()
    v = _f32(sdf)
    if v.dim() != 3 or v.shape[0] != v.shape[1] or v.shape[0] != v.shape[2] or v.shape[0] < 2:
        raise ValueError("sdf must be an (N, N, N) lattice volume")
    dev, N = v.device, int(v.shape[0])
    if dev not in _mc_tables:
        import numpy as np
        from . import meshing
        table_np, ntri_np = meshing.case_table()
        assert table_np.shape[1] <= 16
        t16 = -np.ones((256, 16), np.int8)
        t16[:, :table_np.shape[1]] = table_np
        _mc_tables[dev] = (torch.from_numpy(t16).to(dev), torch.from_numpy(ntri_np.astype(np.int32)).to(dev))
    table, ntri = _mc_tables[dev]
    with _on_device(dev):
        tris = torch.empty(int(cap), 3, 3, device=dev)
        n_tris = torch.empty(1, dtype=torch.int32, device=dev)
        scratch = torch.empty(int(lib.arah_marching_cubes_scratch_bytes(N)), dtype=torch.uint8, device=dev)
        _check(lib.arah_marching_cubes(_ptr(v), C.c_int32(N), C.c_float(float(level)), _ptr(table), _ptr(ntri), _ptr(tris),
                                       C.c_int32(int(cap)), _ptr(n_tris), _ptr(scratch), C.c_size_t(scratch.numel()), _stream()),
               "arah_marching_cubes")
    return tris, n_tris


@_guarded
def skin_lbs_counted(frame, ws, x_hat, n_items, per_item=1):
    """Forward skinning of the first n_items[0] * per_item rows of x_hat (P,3) -- a count that lives on the device -- ->
    x_bar (P,3), zero beyond.  (The vertices of a mesh hip.marching_cubes just extracted, per_item = 3.)"""
    lib = load_library()
    x = _f32(x_hat)
    buf = ws.ensure(1, 1)
    xb = torch.zeros(x.shape[0], 3, device=x.device)
    _check(lib.arah_skin_lbs_counted(C.byref(frame.handle), _ptr(x), C.c_int32(int(x.shape[0])), _ptr(n_items),
                                     C.c_int32(int(per_item)), _ptr(xb), _ptr(buf), C.c_size_t(buf.numel()), _stream()),
           "arah_skin_lbs_counted")
    return xb


def gemv_rows(weight, x, b0=None, b1=None):
    """y = weight @ x + b0 + b1 for ONE vector x (the wide output layers of the hypernetwork, an HBM stream of `weight`);
    no autograd.  weight (R, C) fp32 contiguous, C % 4 == 0."""
    require_gpu()
    lib = load_library()
    w, xv = weight.detach(), _f32(x).reshape(-1)
    if not (w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 2 and w.shape[1] == xv.numel() and w.shape[1] % 4 == 0):
        raise ValueError("gemv_rows: weight (R, C) fp32 contiguous with C % 4 == 0 and x of C elements")
    with _on_device(w.device):
        y = torch.empty(w.shape[0], device=w.device)
        _check(lib.arah_gemv_rows(_ptr(w), C.c_int32(w.shape[0]), C.c_int32(w.shape[1]), _ptr(xv),
                                  _ptr(None if b0 is None else _f32(b0).reshape(-1)),
                                  _ptr(None if b1 is None else _f32(b1).reshape(-1)), _ptr(y), _stream()), "arah_gemv_rows")
    return y


def rasterize(tri_uvz, height, width, z_near=1e-4):
    """tri_uvz (F,3,3) pixel-space corners (u, v, view depth) -> pix_to_face (H,W) int64, -1 where nothing covers."""
    require_gpu()
    lib = load_library()
    tri = _f32(tri_uvz)
    dev = tri.device
    with _on_device(dev):
        zbuf = torch.full((height * width,), -1, dtype=torch.int64, device=dev)   # all bits set
        _check(lib.arah_rasterize(_ptr(tri), C.c_int32(int(tri.shape[0])), C.c_int32(int(height)), C.c_int32(int(width)),
                                  C.c_float(float(z_near)), _ptr(zbuf), _stream()), "arah_rasterize")
    face = (zbuf & 0xffffffff).reshape(height, width)
    return torch.where(zbuf.reshape(height, width) == -1, torch.full_like(face, -1), face)


@_guarded
def skin_lbs(frame, ws, x_hat):
    lib = load_library()
    x = _f32(x_hat)
    n = x.shape[0]
    buf = ws.ensure(1, 1)
    w = torch.empty(n, 24, device=x.device)
    xb = torch.empty(n, 3, device=x.device)
    T = torch.empty(n, 4, 4, device=x.device)
    _check(lib.arah_skin_lbs(C.byref(frame.handle), _ptr(x), C.c_int32(n), _ptr(w), _ptr(xb), _ptr(T), _ptr(buf),
                             C.c_size_t(buf.numel()), _stream()), "arah_skin_lbs")
    return w, xb, T


@_guarded
def skin_jacobian(frame, ws, x_hat):
    lib = load_library()
    x = _f32(x_hat)
    n = x.shape[0]
    buf = ws.ensure(1, 1)
    jac = torch.empty(n, 3, 3, device=x.device)
    _check(lib.arah_skin_jacobian(C.byref(frame.handle), _ptr(x), C.c_int32(n), _ptr(jac), _ptr(buf),
                                  C.c_size_t(buf.numel()), _stream()), "arah_skin_jacobian")
    return jac


@_guarded
def color_eval(frame, ws, x_norm, normal, view, feat):
    lib = load_library()
    x, nr, ft = _f32(x_norm), _f32(normal), _f32(feat)
    vw = _f32(view) if view is not None else None
    n = x.shape[0]
    buf = ws.ensure(1, 1)
    rgb = torch.empty(n, 3, device=x.device)
    _check(lib.arah_color_eval(C.byref(frame.handle), _ptr(x), _ptr(nr), _ptr(vw), _ptr(ft), C.c_int32(n), _ptr(rgb),
                               _ptr(buf), C.c_size_t(buf.numel()), _stream()), "arah_color_eval")
    return rgb


@_guarded
def shade_points(frame, ws, x_norm, T, dirs, cano_view_dirs=True, shade_engine=None):
    """The per-sample half of loop D on the frame's own engine (arah_shade_points): n normalised canonical points with their
    transforms (n,4,4) and ray directions (n,3) -> rgb (n,3), density (n,), sdf (n,), d sdf / d x_norm (n,3)."""
    lib = load_library()
    x, Tm, d = _f32(x_norm), _f32(T), _f32(dirs)
    n = x.shape[0]
    buf = ws.ensure(n, 1)
    rgbs = torch.empty(n, 4, device=x.device)
    sdfn = torch.empty(n, 4, device=x.device)
    eng = default_shade_engine() if shade_engine is None else int(shade_engine)
    _check(lib.arah_shade_points(C.byref(frame.handle), _ptr(x), _ptr(Tm), _ptr(d), C.c_int32(n), C.c_int32(int(bool(cano_view_dirs))),
                                 C.c_int32(eng), _ptr(rgbs), _ptr(sdfn), _ptr(buf), C.c_size_t(buf.numel()), _stream()), "arah_shade_points")
    return rgbs[:, :3], rgbs[:, 3], sdfn[:, 0], sdfn[:, 1:]


@_guarded
def nearest_inverse_lbs(frame, ws, pts):
    lib = load_library()
    p = _f32(pts)
    n = p.shape[0]
    buf = ws.ensure(1, 1)
    idx = torch.empty(n, dtype=torch.int32, device=p.device)
    x0 = torch.empty(n, 3, device=p.device)
    T0 = torch.empty(n, 4, 4, device=p.device)
    _check(lib.arah_nearest_inverse_lbs(C.byref(frame.handle), _ptr(p), C.c_int32(n), _ptr(idx), _ptr(x0), _ptr(T0),
                                        _ptr(buf), C.c_size_t(buf.numel()), _stream()), "arah_nearest_inverse_lbs")
    return idx, x0, T0


@_guarded
def broyden3_lbs(frame, ws, tgt, x0, T0, canon_kernel=None):
    lib = load_library()
    kern = default_canon_kernel() if canon_kernel is None else int(canon_kernel)
    tgt, x0, T0 = _f32(tgt), _f32(x0), _f32(T0)
    n = tgt.shape[0]
    buf = ws.ensure(n, 1)
    x = torch.empty(n, 3, device=tgt.device)
    T = torch.empty(n, 4, 4, device=tgt.device)
    err = torch.empty(n, device=tgt.device)
    conv = torch.empty(n, dtype=torch.uint8, device=tgt.device)
    _check(lib.arah_broyden3_lbs(C.byref(frame.handle), _ptr(tgt), _ptr(x0), _ptr(T0), C.c_int32(n), _ptr(x), _ptr(T),
                                 _ptr(err), _ptr(conv), C.c_int32(kern), _ptr(buf), C.c_size_t(buf.numel()), _stream()),
           "arah_broyden3_lbs")
    return x, T, err, conv.bool()


@_guarded
def joint_root_find(frame, ws, cam_loc, dirs, valid, x0, z0, T0):
    """Loop B on caller-supplied starts: cam_loc (B,3), dirs (N,3), valid (N,) bool/uint8, x0 (N,3) raw canonical,
    z0 (N,), T0 (N,4,4) -> x, z, T, conv."""
    lib = load_library()
    cam, d, x0, z0, T0 = _f32(cam_loc), _f32(dirs), _f32(x0), _f32(z0), _f32(T0)
    v = valid.to(torch.uint8).contiguous()
    n = d.shape[0]
    buf = ws.ensure(n, 1)
    dev = d.device
    x = torch.empty(n, 3, device=dev)
    z = torch.empty(n, device=dev)
    T = torch.empty(n, 4, 4, device=dev)
    conv = torch.empty(n, dtype=torch.uint8, device=dev)
    _check(lib.arah_joint_root_find(C.byref(frame.handle), _ptr(cam), C.c_int32(n // cam.shape[0]), _ptr(d), _ptr(v),
                                    _ptr(x0), _ptr(z0), _ptr(T0), C.c_int32(n), _ptr(x), _ptr(z), _ptr(T), _ptr(conv),
                                    _ptr(buf), C.c_size_t(buf.numel()), _stream()), "arah_joint_root_find")
    return x, z, T, conv.bool()


@_guarded
def trace(frame, ws, cam_loc, dirs, near_far, root_find_all=False):
    """cam_loc (B,3), dirs (B*N,3) flat, near_far (B*N,2) -> x_norm, T, conv, start, end."""
    lib = load_library()
    cam, d, nf = _f32(cam_loc), _f32(dirs), _f32(near_far)
    n = d.shape[0]
    buf = ws.ensure(n, 1)
    dev = d.device
    xn = torch.empty(n, 3, device=dev)
    T = torch.empty(n, 4, 4, device=dev)
    conv = torch.empty(n, dtype=torch.uint8, device=dev)
    start = torch.empty(n, device=dev)
    end = torch.empty(n, device=dev)
    _check(lib.arah_trace(C.byref(frame.handle), _ptr(cam), C.c_int32(n // cam.shape[0]), _ptr(d), _ptr(nf),
                          C.c_int32(n), C.c_int32(int(bool(root_find_all))), _ptr(xn), _ptr(T), _ptr(conv), _ptr(start),
                          _ptr(end), _ptr(buf), C.c_size_t(buf.numel()), _stream()), "arah_trace")
    return xn, T, conv, start, end


@_guarded
def sample_canonicalize(frame, ws, sampling, cam_loc, dirs, near_far, conv, start, end, rand=None):
    """rand: None (eval) or (rand_steps (N,S), rand_near (N,near+1), rand_far (N,far)) uniform draws."""
    lib = load_library()
    r_s = r_n = r_f = None
    if rand is not None:
        r_s, r_n, r_f = [_f32(t) for t in rand]
    cam, d, nf = _f32(cam_loc), _f32(dirs), _f32(near_far)
    n, S = d.shape[0], sampling.n_steps
    buf = ws.ensure(n, S)
    dev = d.device
    z = torch.empty(n, S, device=dev)
    pts = torch.empty(n, S, 3, device=dev)
    T = torch.empty(n, S, 4, 4, device=dev)
    mask = torch.empty(n, S, dtype=torch.uint8, device=dev)
    _check(lib.arah_sample_canonicalize(C.byref(frame.handle), C.byref(sampling.handle), _ptr(cam),
                                        C.c_int32(n // cam.shape[0]), _ptr(d), _ptr(nf), _ptr(conv.contiguous()),
                                        _ptr(_f32(start)), _ptr(_f32(end)), C.c_int32(n), _ptr(r_s), _ptr(r_n), _ptr(r_f),
                                        _ptr(z), _ptr(pts), _ptr(T), _ptr(mask), _ptr(buf), C.c_size_t(buf.numel()),
                                        _stream()),
           "arah_sample_canonicalize")
    return z, pts, T, mask


@_guarded
def shade_composite(frame, ws, sampling, dirs, z, pts, T, mask):
    lib = load_library()
    d = _f32(dirs)
    n, S = d.shape[0], sampling.n_steps
    buf = ws.ensure(n, S)
    dev = d.device
    rgb = torch.empty(n, 3, device=dev)
    acc = torch.empty(n, device=dev)
    vol = torch.empty(n, dtype=torch.uint8, device=dev)
    _check(lib.arah_shade_composite(C.byref(frame.handle), C.byref(sampling.handle), _ptr(d), _ptr(_f32(z)),
                                    _ptr(_f32(pts)), _ptr(_f32(T)), _ptr(mask.contiguous()), C.c_int32(n), _ptr(rgb),
                                    _ptr(acc), _ptr(vol), _ptr(buf), C.c_size_t(buf.numel()), _stream()),
           "arah_shade_composite")
    return rgb, acc, vol


# ------------------------------------------------------------------------------------------------
# loop D with gradients: the pair of entry points behind training.ShadeSamples (torch.autograd.Function)
# ------------------------------------------------------------------------------------------------
KIN_PAD = {COLOR_NO_VIEW_DIR: 272, COLOR_IDR: 304}   # colour input [feat(256) | x(3) | n(3) | PE(view)(27)] padded to 16


def _grouped(groups, n, width, device, inner=None):
    """(groups[, inner], roundup(n, 64), width) floats whose padding rows are zero: the operand streams of products over the sample
    axis that travel together (see tall.gram_grouped).  The kernels write rows [0, n) of each slice."""
    # a step's sample count moves by a few hundred from frame to frame: rounded up to 8192 / 2048 rows the gigabyte-sized groups
    # come in one or two sizes and torch's allocator hands the same blocks out again (rounded to 64 only, every other step of
    # bench.py's training line paid a 35 ms device allocation)
    n_pad = (n + 8191) // 8192 * 8192 if n >= 65536 else ((n + 2047) // 2048 * 2048 if n >= 16384 else (n + 63) // 64 * 64)
    shape = (groups, n_pad, width) if inner is None else (groups, inner, n_pad, width)
    buf = torch.empty(*shape, device=device)
    if n_pad > n:
        buf[..., n:, :].zero_()
    return buf


def _train_in(x, T, view, view_orig, rotate_normal, ray_augm, g_s=None, g_rgb=None):
    t = ArahTrainIn()
    t.n, t.rotate_normal, t.ray_augm = int(x.shape[0]), int(bool(rotate_normal)), int(bool(ray_augm))
    t.x, t.T, t.view, t.view_orig = _ptr(x), _ptr(T), _ptr(view), _ptr(view_orig)
    t.g_s, t.g_rgb = _ptr(g_s), _ptr(g_rgb)
    return t


@_guarded
def shade_train_forward(frame, ws, x, T, view, view_orig, rotate_normal, ray_augm, keep=False):
    """x (P,3) normalised canonical points, T (P,4,4) or None, view / view_orig (P,3) -> sdf (P,), rgb (P,3).
    keep=True: also the hand-over for shade_train_backward -- dict(cin, c, rgb4): the colour MLP's input and hidden
    activations as (P, width) streams and the padded rgb; the backward then skips the normal sweep and the colour MLP."""
    lib = load_library()
    x, view = _f32(x), _f32(view)
    T = _f32(T) if (T is not None and rotate_normal) else None
    vo = _f32(view_orig) if (view_orig is not None and ray_augm) else None
    n = x.shape[0]
    buf = ws.ensure(1, 1)
    sdf = torch.empty(n, device=x.device)
    rgb4 = torch.empty(n, 4, device=x.device)
    tin = _train_in(x, T, view, vo, rotate_normal, ray_augm)
    kept = None
    if keep:
        # the 256-wide activations sit next to each other, padded to a multiple of 64 rows with zeros: the weight-gradient
        # products of the backward then run as ONE batched split-K product per group (tall.gram_grouped), all views
        cc = _grouped(4, n, 256, x.device)
        cinp, c2p = _grouped(1, n, KIN_PAD[frame.color_mode], x.device)[0], _grouped(1, n, 128, x.device)[0]
        kept = {"cin": cinp[:n], "c": [cc[0, :n], cc[2, :n], c2p[:n], cc[1, :n], cc[3, :n]], "rgb4": rgb4, "cc": cc,
                "cin_pad": cinp, "c2_pad": c2p}
        tin.tap_cin = _ptr(kept["cin"])
        for i, t in enumerate(kept["c"]):
            tin.tap_c[i] = _ptr(t).value
    _check(lib.arah_shade_train_forward(C.byref(frame.handle), C.byref(tin), _ptr(sdf), _ptr(rgb4), _ptr(buf),
                                        C.c_size_t(buf.numel()), _stream()), "arah_shade_train_forward")
    return (sdf, rgb4[:, :3], kept) if keep else (sdf, rgb4[:, :3])


def _sdf_streams(n, dev):
    """Operand streams of the SIREN's weight gradients, grouped for tall.gram_grouped: `avv` (6, 2, n_pad, 256) holds adj v_k and
    adj vd_k of layer k next to each other, `hh` (5, 2, n_pad, 256) their partners h_k and hd_k (k = 1..5), `h0` (2, n_pad, 4) the
    first layer's h_0 = x and hd_0 = nt.  dW_k = adj(v_k)^T h_{k-1} + adj(vd_k)^T hd_{k-1} is then ONE product over 2 n_pad rows."""
    avv, hh, h0 = _grouped(6, n, 256, dev, inner=2), _grouped(5, n, 256, dev, inner=2), _grouped(2, n, 4, dev)
    return {"h": [h0[0, :n]] + [hh[k, 0, :n] for k in range(5)],
            "hd": [h0[1, :n]] + [hh[k, 1, :n] for k in range(5)] + [torch.empty(n, 256, device=dev)],
            "av": [avv[k, 0, :n] for k in range(6)], "avd": [avv[k, 1, :n] for k in range(6)], "avv": avv, "hh": hh, "h0": h0}


@_guarded
def sdf_normal_forward(frame, ws, x):
    """The regulariser queries (IDR:104-128): x (P,3) normalised canonical points -> sdf (P,) in normalised units and its
    gradient d sdf / d x (P,3), by the training kernel without its colour half (ArahTrainIn.geom_only)."""
    lib = load_library()
    x = _f32(x)
    n = x.shape[0]
    buf = ws.ensure(1, 1)
    sdf = torch.empty(n, device=x.device)
    n4 = torch.empty(n, 4, device=x.device)
    tin = _train_in(x, None, None, None, False, False)
    tin.geom_only = 1
    _check(lib.arah_shade_train_forward(C.byref(frame.handle), C.byref(tin), _ptr(sdf), _ptr(n4), _ptr(buf),
                                        C.c_size_t(buf.numel()), _stream()), "arah_shade_train_forward")
    return sdf, n4[:, :3]


@_guarded
def sdf_normal_backward(frame, ws, x, g_s, g_n):
    """dL/dx (P,3) and the operand streams of the SDF weight gradients (h, hd, av, avd, the feature h_6 as `feat`,
    film_freq, film_phase) for upstream gradients g_s (P,) on the value and g_n (P,3) on the normal."""
    lib = load_library()
    x, g_s, g_n = _f32(x), _f32(g_s), _f32(g_n)
    n, dev = x.shape[0], x.device
    buf = ws.ensure(1, 1)
    E = lambda *shape: torch.empty(*shape, device=dev)
    st = {"gx4": E(n, 4), "film_freq": E(6, 256), "film_phase": E(6, 256), "feat": E(n, 256)}
    st.update(_sdf_streams(n, dev))
    g = ArahTrainGrads()
    for k in ("gx4", "film_freq", "film_phase"):
        setattr(g, k, _ptr(st[k]))
    for k in ("h", "hd", "av", "avd"):
        arr = getattr(g, k)
        for i, t in enumerate(st[k]):
            arr[i] = _ptr(t).value
    g.c[0] = _ptr(st["feat"]).value
    nslab = lib.arah_shade_train_slab_bytes()
    if getattr(ws, "train_slab", None) is None or ws.train_slab.numel() < nslab:
        ws.train_slab = torch.empty(nslab, dtype=torch.uint8, device=dev)
    tin = _train_in(x, None, None, None, False, False, g_s, g_n)
    tin.geom_only = 1
    _check(lib.arah_shade_train_backward(C.byref(frame.handle), C.byref(tin), C.byref(g), _ptr(ws.train_slab),
                                         C.c_size_t(ws.train_slab.numel()), _ptr(buf), C.c_size_t(buf.numel()),
                                         _stream()), "arah_shade_train_backward")
    return st


@_guarded
def shade_train_backward(frame, ws, x, T, view, view_orig, rotate_normal, ray_augm, g_s, g_rgb, kept=None):
    """Recomputes the forward and returns the per-sample gradient dL/dx (P,3), the FiLM gradients (6,256) x 2 and the
    operand streams of the weight-gradient GEMMs (dict of dense (P, width) tensors, see ArahTrainGrads)."""
    lib = load_library()
    x, view, g_s, g_rgb = _f32(x), _f32(view), _f32(g_s), _f32(g_rgb)
    T = _f32(T) if (T is not None and rotate_normal) else None
    vo = _f32(view_orig) if (view_orig is not None and ray_augm) else None
    n, dev = x.shape[0], x.device
    buf = ws.ensure(1, 1)
    kin = KIN_PAD[frame.color_mode]
    E = lambda *shape: torch.empty(*shape, device=dev)
    if kept:
        cc, cinp, c2p = kept["cc"], kept["cin_pad"], kept["c2_pad"]
    else:
        cc, cinp, c2p = _grouped(4, n, 256, dev), _grouped(1, n, kin, dev)[0], _grouped(1, n, 128, dev)[0]
    dd = _grouped(4, n, 256, dev)     # delta_1, delta_4, delta_0, delta_3: the order of their partners in `cc` (c_1, c_4, c_2, c_5)
    d2p = _grouped(1, n, 128, dev)[0]
    st = {"sdf": E(n), "rgb4": E(n, 4), "gx4": E(n, 4), "film_freq": E(6, 256), "film_phase": E(6, 256),
          "cin": cinp[:n], "c": kept["c"] if kept else [cc[0, :n], cc[2, :n], c2p[:n], cc[1, :n], cc[3, :n]],
          "d": [dd[2, :n], dd[0, :n], d2p[:n], dd[3, :n], dd[1, :n], E(n, 4)], "cc": cc, "dd": dd,
          "cin_pad": cinp, "c2_pad": c2p, "d2_pad": d2p}   # the *_pad views: whole buffers, padding rows zero (no remainder product)
    st.update(_sdf_streams(n, dev))
    g = ArahTrainGrads()
    for k in ("sdf", "rgb4", "gx4", "film_freq", "film_phase", "cin"):
        setattr(g, k, _ptr(st[k]))
    for k in ("h", "hd", "av", "avd", "c", "d"):
        arr = getattr(g, k)
        for i, t in enumerate(st[k]):
            arr[i] = _ptr(t).value
    nslab = lib.arah_shade_train_slab_bytes()
    if getattr(ws, "train_slab", None) is None or ws.train_slab.numel() < nslab:
        ws.train_slab = torch.empty(nslab, dtype=torch.uint8, device=dev)
    tin = _train_in(x, T, view, vo, rotate_normal, ray_augm, g_s, g_rgb)
    if kept:
        tin.tap_cin = _ptr(kept["cin"])
        for i, t in enumerate(kept["c"]):
            tin.tap_c[i] = _ptr(t).value
        tin.fwd_rgb4 = _ptr(kept["rgb4"])
    _check(lib.arah_shade_train_backward(C.byref(frame.handle), C.byref(tin), C.byref(g), _ptr(ws.train_slab),
                                         C.c_size_t(ws.train_slab.numel()), _ptr(buf), C.c_size_t(buf.numel()),
                                         _stream()), "arah_shade_train_backward")
    return st


@_guarded
def composite_train_forward(lengths, offsets, sdf, rgb, z, inv_beta, n_steps, render_last_pt):
    """lengths (R,) int32, offsets (R,) int64 into the compacted per-sample arrays sdf (P,) [metres], rgb (P,3), z (P,);
    inv_beta (1,) -> rgb_map (R,3), acc (R,) = clip(sum of weights, 0, 1)."""
    lib = load_library()
    R, dev = int(lengths.shape[0]), sdf.device
    out_rgb, out_acc = torch.empty(R, 3, device=dev), torch.empty(R, device=dev)
    _check(lib.arah_composite_train_forward(C.c_int32(R), C.c_int32(int(n_steps)), C.c_int32(int(bool(render_last_pt))),
                                            _ptr(lengths), _ptr(offsets), _ptr(sdf), _ptr(rgb), _ptr(z), _ptr(inv_beta),
                                            _ptr(out_rgb), _ptr(out_acc), _stream()), "arah_composite_train_forward")
    return out_rgb, out_acc


@_guarded
def composite_train_backward(lengths, offsets, sdf, rgb, z, inv_beta, n_steps, render_last_pt, g_map, g_acc):
    """-> dL/d sdf (P,), dL/d rgb (P,3), dL/d inv_beta (1,)"""
    lib = load_library()
    R, dev = int(lengths.shape[0]), sdf.device
    g_sdf, g_rgb, g_ib = torch.empty_like(sdf), torch.empty_like(rgb), torch.empty(1, device=dev)
    _check(lib.arah_composite_train_backward(C.c_int32(R), C.c_int32(int(n_steps)), C.c_int32(int(bool(render_last_pt))),
                                             _ptr(lengths), _ptr(offsets), _ptr(sdf), _ptr(rgb), _ptr(z), _ptr(inv_beta),
                                             _ptr(_f32(g_map)), _ptr(_f32(g_acc)), _ptr(g_sdf), _ptr(g_rgb), _ptr(g_ib),
                                             _stream()), "arah_composite_train_backward")
    return g_sdf, g_rgb, g_ib


def mesh_query(verts, faces, pts):
    """Closest point of a triangle mesh and containment for every query point (training samplers, zju_mocap.py:461-543).
    verts (V,3) float32, faces (F,3) int32, pts (P,3) float32 or float64 on one GPU ->
    d2 (P,) float64, face (P,) int32, closest (P,3) float64, bary (P,3) float64, inside (P,) bool."""
    lib = load_library()
    dev = _same_device(verts, faces, pts)
    if verts.dtype != torch.float32 or faces.dtype != torch.int32 or pts.dtype not in (torch.float32, torch.float64):
        raise ValueError("verts float32, faces int32, pts float32 / float64 required")
    verts, faces, pts = verts.contiguous(), faces.contiguous(), pts.contiguous()
    P = pts.shape[0]
    d2 = torch.empty(P, dtype=torch.float64, device=dev)
    face = torch.empty(P, dtype=torch.int32, device=dev)
    closest = torch.empty(P, 3, dtype=torch.float64, device=dev)
    bary = torch.empty(P, 3, dtype=torch.float64, device=dev)
    inside = torch.empty(P, dtype=torch.uint8, device=dev)
    with _on_device(dev):
        scratch = torch.empty(lib.arah_mesh_query_scratch_bytes(), dtype=torch.uint8, device=dev)
        _check(lib.arah_mesh_query(_ptr(verts), C.c_int32(verts.shape[0]), _ptr(faces), C.c_int32(faces.shape[0]), _ptr(pts),
                                   C.c_int32(1 if pts.dtype == torch.float64 else 0), C.c_int32(P), _ptr(d2), _ptr(face),
                                   _ptr(closest), _ptr(bary), _ptr(inside), _ptr(scratch), _stream()), "arah_mesh_query")
    return d2, face, closest, bary, inside.bool()


def gram_skinny(a, b):
    """a^T b for a (P, m <= 4) and b (P, n), row strides free (column slices of wider streams are fine): (m, n)."""
    lib = load_library()
    dev = _same_device(a, b)
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.stride(1) != 1 or b.stride(1) != 1:
        raise ValueError("float32 operands with unit column stride required")
    P, m = a.shape
    n = b.shape[1]
    if P == 0:
        return torch.zeros(m, n, device=dev)
    with _on_device(dev):
        blocks = lib.arah_gram_skinny_blocks(C.c_int32(P))
        partial = torch.empty(blocks, m, n, device=dev)
        _check(lib.arah_gram_skinny(C.c_void_p(a.data_ptr()), C.c_int32(a.stride(0)), C.c_int32(m),
                                    C.c_void_p(b.data_ptr()), C.c_int32(b.stride(0)),
                                    C.c_int32(n), C.c_int32(P), _ptr(partial), _stream()), "arah_gram_skinny")
    return partial.sum(0)


def colsum(a, scale=None):
    """y[c] = sum_r scale[r] a[r, c] for a (R, n) fp32 with unit column stride (row stride free), scale (R,) or None: (n,).
    One pass at HBM speed with a fixed summation order (arah_colsum)."""
    lib = load_library()
    dev = _same_device(a, scale)
    if a.dtype != torch.float32 or a.dim() != 2 or a.stride(1) != 1:
        raise ValueError("colsum: (R, n) float32 with unit column stride required")
    R, n = a.shape
    sc = None if scale is None else _f32(scale).reshape(-1)
    if sc is not None and sc.numel() != R:
        raise ValueError("colsum: one scale per row")
    with _on_device(dev):
        y = torch.empty(n, device=dev)
        partial = torch.empty(max(1, int(lib.arah_colsum_blocks(C.c_int64(R)))), n, device=dev)
        _check(lib.arah_colsum(C.c_void_p(a.data_ptr()), C.c_int64(a.stride(0) if R > 1 else n), C.c_int32(n), C.c_int64(R),
                               _ptr(sc), _ptr(partial), _ptr(y), _stream()), "arah_colsum")
    return y


def hsoftmax_train_forward(logits, scale):
    """(n, 25) raw logits -> (n, 24) weights = hierarchical_softmax(scale * logits) (arah_hsoftmax_train_forward); no autograd."""
    lib = load_library()
    x = _f32(logits)
    dev = _same_device(x)
    with _on_device(dev):
        w = torch.empty(x.shape[0], 24, device=dev)
        _check(lib.arah_hsoftmax_train_forward(_ptr(x), C.c_int32(x.shape[0]), C.c_float(float(scale)), _ptr(w), _stream()),
               "arah_hsoftmax_train_forward")
    return w


def hsoftmax_train_backward(logits, scale, g_w):
    """d L / d logits (n, 25) for upstream gradients g_w (n, 24) on hierarchical_softmax(scale * logits)."""
    lib = load_library()
    x, g = _f32(logits), _f32(g_w)
    dev = _same_device(x, g)
    with _on_device(dev):
        gx = torch.empty_like(x)
        _check(lib.arah_hsoftmax_train_backward(_ptr(x), C.c_int32(x.shape[0]), C.c_float(float(scale)), _ptr(g), _ptr(gx),
                                                _stream()), "arah_hsoftmax_train_backward")
    return gx


def _parents_array(parents):
    arr = (C.c_int32 * len(parents))(*[int(p) for p in parents])
    return arr


def pose_tree_forward(own, glob, W1, b1, W2, b2, parents):
    """own (J,13), glob (6,), stacked per-joint parameters W1 (J,19,19), b1 (J,19), W2 (J,6,19), b2 (J,6), parents (list, -1 = root)
    -> feats (J,6), hidden (J,19) (arah_pose_tree_forward); no autograd."""
    lib = load_library()
    ts = [_f32(t) for t in (own, glob, W1, b1, W2, b2)]
    dev = _same_device(*ts)
    J = ts[0].shape[0]
    if ts[0].shape != (J, 13) or ts[1].numel() != 6 or ts[2].shape != (J, 19, 19) or ts[3].shape != (J, 19) \
            or ts[4].shape != (J, 6, 19) or ts[5].shape != (J, 6) or len(parents) != J:
        raise ValueError("pose_tree_forward: shapes of a 13 + 6 -> 19 -> 6 tree encoder expected")
    with _on_device(dev):
        feats, hidden = torch.empty(J, 6, device=dev), torch.empty(J, 19, device=dev)
        _check(lib.arah_pose_tree_forward(*[_ptr(t) for t in ts], _parents_array(parents), C.c_int32(J), _ptr(feats), _ptr(hidden),
                                          _stream()), "arah_pose_tree_forward")
    return feats, hidden


def pose_tree_backward(own, glob, W1, W2, parents, feats, hidden, g_feats):
    """-> (gW1, gb1, gW2, gb2, g_glob) for upstream gradients g_feats (J,6) (arah_pose_tree_backward)."""
    lib = load_library()
    own, glob, W1, W2, feats, hidden, g = [_f32(t) for t in (own, glob, W1, W2, feats, hidden, g_feats)]
    dev = _same_device(own, glob, W1, W2, feats, hidden, g)
    J = own.shape[0]
    with _on_device(dev):
        gW1, gb1 = torch.empty(J, 19, 19, device=dev), torch.empty(J, 19, device=dev)
        gW2, gb2, gg = torch.empty(J, 6, 19, device=dev), torch.empty(J, 6, device=dev), torch.empty(6, device=dev)
        _check(lib.arah_pose_tree_backward(_ptr(own), _ptr(glob), _ptr(W1), _ptr(W2), _parents_array(parents), C.c_int32(J),
                                           _ptr(feats), _ptr(hidden), _ptr(g), _ptr(gW1), _ptr(gb1), _ptr(gW2), _ptr(gb2), _ptr(gg),
                                           _stream()), "arah_pose_tree_backward")
    return gW1, gb1, gW2, gb2, gg


def inverse3x3(m, scale=1.0):
    """(scale * m)^-1 for m (P, 3, 3) fp32 by cofactors (arah_inverse3x3); no autograd."""
    lib = load_library()
    mm = _f32(m)
    dev = _same_device(mm)
    if mm.dim() != 3 or mm.shape[1:] != (3, 3):
        raise ValueError("inverse3x3: (P, 3, 3) required")
    with _on_device(dev):
        out = torch.empty_like(mm)
        _check(lib.arah_inverse3x3(_ptr(mm), C.c_int32(mm.shape[0]), C.c_float(float(scale)), _ptr(out), _stream()),
               "arah_inverse3x3")
    return out


@_guarded
def render(frame, ws, sampling, cam_loc, dirs, near_far, pose34, tiered=False):
    """Whole eval forward. pose34: DEVICE (3,4) world->camera (no host copy, no stream drain).
    tiered: build the frame's occupancy bitmap first and let arah_render skip the samples it certifies (csrc/tier.hpp;
    lazy shading only -- with full_shading the flag is ignored).
    Returns rgb, points_cam, vol_mask, acc, dists, conv."""
    lib = load_library()
    cam, d, nf = _f32(cam_loc), _f32(dirs), _f32(near_far)
    n, S = d.shape[0], sampling.n_steps
    buf = ws.ensure(n, S)
    cfg = sampling.handle
    if tiered and not cfg.full_shading:
        occ = ws.occupancy(frame)
        buf = ws.buf
        cfg = ArahSampling.from_buffer_copy(sampling.handle)   # the bitmap is this frame's: a private copy of the call's struct
        cfg.occupancy = occ.data_ptr()
    dev = d.device
    rgb = torch.empty(n, 3, device=dev)
    pcam = torch.empty(n, 3, device=dev)
    vol = torch.empty(n, dtype=torch.uint8, device=dev)
    acc = torch.empty(n, device=dev)
    dists = torch.empty(n, device=dev)
    conv = torch.empty(n, dtype=torch.uint8, device=dev)
    d_pose = _f32(pose34).reshape(-1)[:12].contiguous()
    _check(lib.arah_render(C.byref(frame.handle), C.byref(cfg), _ptr(cam), C.c_int32(n // cam.shape[0]),
                           _ptr(d), _ptr(nf), _ptr(d_pose), C.c_int32(n), _ptr(rgb), _ptr(pcam), _ptr(vol), _ptr(acc),
                           _ptr(dists), _ptr(conv), _ptr(buf), C.c_size_t(buf.numel()), _stream()), "arah_render")
    return rgb, pcam, vol, acc, dists, conv
