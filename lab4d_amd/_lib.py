"""Build + load liblab4d_hip.so (the C-ABI in include/lab4d_hip.h) through ctypes.

No torch types cross the boundary: tensors are passed as raw device pointers, the stream as the
hipStream_t handle of torch's current stream.  There is NO fallback: if the shared library is
missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
SO_PATH = os.environ.get("LAB4D_SO_PATH", os.path.join(HERE, "liblab4d_hip.so"))  # override: kernel experiments only
BUILD_DIR = os.environ.get("LAB4D_BUILD_DIR", os.path.join(HERE, "build"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
          "-Wno-unused-value", "-Wno-pass-failed"] + os.environ.get("LAB4D_HIPCC_EXTRA", "").split()


# the chain-kernel instantiation units only: LLVM's "unclustered high register pressure" rescheduling stage sinks the LDS reads of the shared
# weight groups next to the MFMAs that use them (exposed LDS latency in the one-wave-per-SIMD kernels); measured with it off, together with
# the conditional accumulator fence (mlp_kernels.hpp acc_fence_if): 739 vs 756 ms per step (profiles/r04_flag_variants.json)
MLP_INST_FLAGS = ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(src, dst):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 every csrc/*.hip, link liblab4d_hip.so in-tree."""
    extra = os.environ.get("LAB4D_HIPCC_EXTRA", "")
    if "amdgpu-mfma-vgpr-form" in extra and "LAB4D_MFMA_VGPR_FORM" not in extra:
        # csrc/mlp_kernels.hpp acc_fence_if: the one-wave-per-SIMD chain kernels drop the MFMA -> inline-asm hazard fence because their accumulators live in
        # AGPRs (a compiler-visible v_accvgpr_read sits in between); with the MFMA results forced into VGPRs that is no longer true
        raise RuntimeError("LAB4D_HIPCC_EXTRA passes -amdgpu-mfma-vgpr-form without -DLAB4D_MFMA_VGPR_FORM: the accumulator hazard fence would be dropped")
    os.makedirs(BUILD_DIR, exist_ok=True)
    # The compile flags are part of the staleness key: an experiment build (-DLAB4D_ABL_* ...) in this BUILD_DIR followed by a
    # default build must not leave objects of the other flag set behind -- rebuilds are per object and mtime-based, and lab4d_build_flags()
    # only reports what runtime.hip's own object saw.  A stamp of the full flag set lives beside the objects; when it differs, everything is rebuilt.
    stamp_path = os.path.join(BUILD_DIR, "flags.stamp")
    stamp = "\n".join([HIPCC] + CFLAGS + ["--inst--"] + (MLP_INST_FLAGS if os.environ.get("LAB4D_NO_INST_FLAGS", "0") != "1" else [])) + "\n"
    try:
        with open(stamp_path) as fh:
            stale_flags = fh.read() != stamp
    except OSError:
        stale_flags = True
    if stale_flags:
        force = True
        if os.path.exists(stamp_path):
            os.remove(stamp_path)  # (written again only after every object of the new flag set exists)
    jobs = []
    objs = []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(BUILD_DIR, f[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        # (round 5: the instantiation flags used to be dropped silently whenever LAB4D_HIPCC_EXTRA was set -- an experiment build then differed from
        # the shipped one in more than its -D; LAB4D_NO_INST_FLAGS=1 is the explicit way to build without them)
        inst = os.path.basename(src).startswith("mlp_inst_") and os.environ.get("LAB4D_NO_INST_FLAGS", "0") != "1"
        cmd = [HIPCC] + CFLAGS + (MLP_INST_FLAGS if inst else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose:
            print("[lab4d_amd] compiled", os.path.basename(src), file=sys.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if stale_flags:
        with open(stamp_path, "w") as fh:
            fh.write(stamp)
    if jobs or not os.path.exists(SO_PATH):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
        if verbose:
            print("[lab4d_amd] linked", SO_PATH, file=sys.stderr)
    return SO_PATH


_LIB = None
vp, ci, cu, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_float


class FieldList(ctypes.Structure):
    _fields_ = [("n_fields", ci), ("fields", vp * 16), ("channels", ci * 16), ("modes", ci * 16)]


class FieldGrads(ctypes.Structure):
    _fields_ = [("n_fields", ci), ("fields", vp * 16)]


# name -> argtypes; the single source of truth for the exported symbols (tests check that every
# symbol declared in include/*.h is listed here and resolves in the .so)
SIGNATURES = {
    "lab4d_quaternion_mul_forward": [vp, vp, vp, cu, cu, cu, ci, vp],
    "lab4d_quaternion_mul_backward": [vp, cu, cu, cu, vp, vp, vp, vp, ci, vp],
    "lab4d_quaternion_mul_backward_backward": [vp, vp, cu, cu, cu, vp, vp, vp, vp, vp, vp, ci, vp],
    "lab4d_quaternion_conjugate": [vp, cu, vp, ci, vp],
    "lab4d_mat3x3_det_forward": [vp, vp, cu, ci, vp],
    "lab4d_mat3x3_scale_adjoint_forward": [vp, vp, vp, cu, ci, vp],
    "lab4d_mat3x3_inv_forward": [vp, vp, vp, cu, ci, vp],
    "lab4d_mat3x3_inv_backward": [vp, vp, vp, cu, ci, vp],
    "lab4d_ray_samples_forward": [vp] * 6 + [ci] * 3 + [vp] * 6 + [vp],
    "lab4d_ray_samples_backward": [vp] * 6 + [ci] * 3 + [vp] * 5 + [vp] * 3 + [vp],
    "lab4d_sample_pdf": [vp, vp, ci, ci, ci, cf, vp, vp, vp],
    "lab4d_sample_pdf_u": [vp, vp, vp, ci, ci, ci, cf, vp, vp, vp],
    "lab4d_sort_depth": [vp, ci, vp, ci, ci, vp, vp],
    "lab4d_composite_forward": [vp, vp, ctypes.POINTER(FieldList), vp, vp, vp, ci, ci] + [vp] * 8 + [vp],
    "lab4d_composite_backward": [vp, vp, ctypes.POINTER(FieldList), vp, vp, vp, ci, ci] + [vp] * 5 + [vp, vp,
                                 ctypes.POINTER(FieldGrads), vp, vp, vp] + [vp],
}


INT64_RETURNS = ("lab4d_mlp_packed_bytes", "lab4d_compact_work_ints", "lab4d_skin_blend_backward_workspace_floats")  # host-only size queries


def register(name, argtypes):
    SIGNATURES[name] = argtypes
    if _LIB is not None:
        fn = getattr(_LIB, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name in INT64_RETURNS else ci


def lib():
    """The loaded shared library.  Raises if it has not been built (no silent fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "liblab4d_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "lab4d_amd has no CPU or eager-PyTorch fallback." % SO_PATH)
        _LIB = ctypes.CDLL(SO_PATH)
        _LIB.lab4d_last_error.restype = ctypes.c_char_p
        _LIB.lab4d_arch.restype = ctypes.c_char_p
        _LIB.lab4d_build_flags.restype = ctypes.c_char_p
        flags = _LIB.lab4d_build_flags().decode().split()
        if flags and os.environ.get("LAB4D_ALLOW_EXPERIMENT_BUILD", "0") != "1":
            # kernel-experiment builds (tools/build_variants.sh) are loaded through LAB4D_SO_PATH by the timing tools only, which set the override
            raise RuntimeError("%s was compiled with kernel-experiment macros %s (most of them give wrong results); rebuild without them, or set "
                               "LAB4D_ALLOW_EXPERIMENT_BUILD=1 for a timing experiment" % (SO_PATH, flags))
        for name, at in SIGNATURES.items():
            fn = getattr(_LIB, name)
            fn.argtypes = at
            fn.restype = ctypes.c_int64 if name in INT64_RETURNS else ci
    return _LIB if PROF is None else _ProfiledLib(_LIB)


class _ProfiledLib:
    """While per-kernel profiling is on (PROF is a dict): every entry point that is not already inside a `timed` block is timed under its own
    name, so the per-kernel table of the bench line accounts for ALL of the library's launches, not only the ones with a work model."""

    def __init__(self, lib_):
        self._lib = lib_

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if PROF is None or _TIMED_DEPTH > 0 or not name.startswith("lab4d_") or name in ("lab4d_last_error", "lab4d_arch", "lab4d_build_flags", "lab4d_mlp_fused_backward_supported", "lab4d_mlp_describe", "lab4d_mlp_packed_bytes", "lab4d_compact_work_ints", "lab4d_global_match_workspace_floats", "lab4d_skin_blend_backward_workspace_floats"):  # host-only
            return fn

        def call(*a):
            with timed(name[6:]):
                return fn(*a)
        return call


# LAB4D_NANCHECK=1 (diagnostic): every tensor whose pointer is handed to the library is remembered, and after each entry point returns
# the device is synchronised and the floating-point ones are scanned for non-finite values -- the first report names the kernel that
# produced (or was fed) them.  Together with torch.utils.deterministic.fill_uninitialized_memory (torch.empty -> NaN) this finds reads
# of uninitialised buffers.  Off by default: one attribute test per pointer.
NANCHECK = os.environ.get("LAB4D_NANCHECK", "0") == "1"
NANCHECK_IGNORE = set(os.environ.get("LAB4D_NANCHECK_IGNORE", "").split(","))
_SEEN = []
_REPORTED = set()


def _nancheck(what):
    torch.cuda.synchronize()
    for i, t in enumerate(_SEEN):
        if t.dtype.is_floating_point and t.numel() and what not in NANCHECK_IGNORE:
            flat = t.reshape(-1)
            n, first = 0, -1
            for o in range(0, flat.numel(), 1 << 28):  # pieces: index arithmetic of nonzero() overflows on > 2^31 elements
                piece = flat[o:o + (1 << 28)]
                bad = ~(piece.abs() < 1e15)  # non-finite, or a magnitude no quantity of this renderer reaches (uninitialised memory)
                k = int(bad.sum())
                if k and first < 0:
                    first = o + int(bad.to(torch.uint8).argmax())
                n += k
            if n and n < t.numel() and (what, i) not in _REPORTED:  # a buffer that is ALL poison has simply not been written yet
                _REPORTED.add((what, i))
                print("[nancheck] %s: tensor #%d %s %s has %d non-finite / huge values of %d (first at flat index %d: %s)"
                      % (what, i, tuple(t.shape), str(t.dtype).replace("torch.", ""), n, t.numel(), first, float(flat[first])), file=sys.stderr)
    _SEEN.clear()


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().lab4d_last_error().decode()))
    if NANCHECK:
        _nancheck(what)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if NANCHECK:
        _SEEN.append(t)
    return ctypes.c_void_p(t.data_ptr())


def dp(t):
    """Raw device address of a tensor, for pointer fields of the argument structs."""
    if NANCHECK:
        _SEEN.append(t)
    return t.data_ptr()


def ptr_at(t):
    """Device pointer of the first element of a (possibly non-contiguous) view; the caller passes the row stride."""
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.float64: 2}


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("lab4d_amd ops need device (HIP) tensors; got a %s tensor -- there is no CPU path" % t.device)
        if not t.is_contiguous():
            raise RuntimeError("lab4d_amd ops need contiguous tensors")


# --------------------------------------------------------------------------------------------------
# optional per-kernel-family timing with HIP events on the launch stream (used by bench.py's roofline)
# --------------------------------------------------------------------------------------------------
_TIMED_DEPTH = 0
PROF = None  # set to {} to enable: name -> list of (start_event, end_event, work)


class timed:
    """`with timed("mlp_fwd_base", flops): launch...` records HIP events around the launches on torch's
    current stream (the stream the kernels are launched on)."""

    def __init__(self, name, work=0.0):
        self.name, self.work = name, work

    def __enter__(self):
        global _TIMED_DEPTH
        _TIMED_DEPTH += 1
        self.on = PROF is not None and _TIMED_DEPTH == 1  # nested blocks belong to the outer one
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        global _TIMED_DEPTH
        _TIMED_DEPTH -= 1
        if self.on and PROF is not None:
            self.e.record()
            PROF.setdefault(self.name, []).append((self.s, self.e, self.work))
        return False


def prof_summary():
    """name -> (launches, total_ms, total_flops, total_bytes). Call after torch.cuda.synchronize()."""
    out = {}
    for k, v in (PROF or {}).items():
        fl = sum((w[0] if isinstance(w, tuple) else w) for _, _, w in v)
        by = sum((w[1] if isinstance(w, tuple) else 0.0) for _, _, w in v)
        out[k] = (len(v), sum(s.elapsed_time(e) for s, e, _ in v), fl, by)
    return out
