"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol that include/*.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

from lab4d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for f in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not f.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", f)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(lab4d_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def so():
    _lib.build(verbose=False)
    return ctypes.CDLL(_lib.SO_PATH)


def test_library_exports_every_declared_symbol(so):
    names = declared_symbols()
    assert len(names) >= 29
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing


def test_python_signatures_cover_the_header():
    import lab4d_amd.deformable  # noqa: F401  (registers the mlp / skinning / gauss-density signatures)
    import lab4d_amd.mlp  # noqa: F401
    import lab4d_amd.hashgrid  # noqa: F401
    import lab4d_amd.ingest  # noqa: F401
    import lab4d_amd.multifields  # noqa: F401
    import lab4d_amd.optim  # noqa: F401
    import lab4d_amd.pose  # noqa: F401
    import lab4d_amd.warping  # noqa: F401
    sig = set(_lib.SIGNATURES)
    hdr = set(declared_symbols()) - {"lab4d_last_error", "lab4d_version", "lab4d_arch", "lab4d_build_flags"}
    assert hdr <= sig, sorted(hdr - sig)


def test_library_targets_gfx950(so):
    so.lab4d_arch.restype = ctypes.c_char_p
    assert so.lab4d_arch() == b"gfx950"
    assert so.lab4d_version() >= 1


def test_shipped_library_has_no_experiment_macros(so):
    """The 40-odd LAB4D_ABL_* / LAB4D_WSABL_* / LAB4D_WS_TRACE ... switches in csrc/ are timing experiments (most give wrong results): the library the
    product loads must have been compiled with none of them, and lab4d_build_flags() must know every such macro the sources test."""
    import re
    so.lab4d_build_flags.restype = ctypes.c_char_p
    assert so.lab4d_build_flags() == b"", so.lab4d_build_flags()
    csrc = os.path.join(ROOT, "lab4d_amd", "csrc")
    tested = set()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")):
            src = open(os.path.join(csrc, f)).read()
            tested |= set(re.findall(r"#\s*ifn?def\s+(LAB4D_[A-Z0-9_]+)", src)) | set(re.findall(r"defined\((LAB4D_[A-Z0-9_]+)\)", src))
    tested -= {"LAB4D_HIP_H"}
    runtime = open(os.path.join(csrc, "runtime.hip")).read()
    reported = set(re.findall(r"#ifdef (LAB4D_[A-Z0-9_]+)", runtime))
    assert tested <= reported, sorted(tested - reported)


def test_bad_arguments_are_rejected_without_a_gpu(so):
    # argument validation happens before any launch, so it is testable on CPU
    so.lab4d_last_error.restype = ctypes.c_char_p
    rc = so.lab4d_quaternion_mul_forward(None, None, None, 4, 4, 4, 0, None)
    assert rc == -1 and b"null" in so.lab4d_last_error()
    buf = ctypes.create_string_buffer(64)
    rc = so.lab4d_quaternion_mul_forward(buf, buf, buf, ctypes.c_uint32(1), ctypes.c_uint32(5), ctypes.c_uint32(4), 0, None)
    assert rc == -1 and b"3 or 4" in so.lab4d_last_error()
    rc = so.lab4d_quaternion_conjugate(buf, ctypes.c_uint32(1), buf, 9, None)
    assert rc == -1 and b"dtype" in so.lab4d_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from lab4d_amd import quaternion
    with pytest.raises(RuntimeError, match="no CPU path"):
        quaternion.quaternion_mul(torch.randn(3, 4), torch.randn(3, 4))


def test_fk_arguments_are_validated(so):
    so.lab4d_last_error.restype = ctypes.c_char_p
    buf = ctypes.create_string_buffer(64)
    rc = so.lab4d_fk_forward(buf, buf, None, buf, buf, 1, 33, 0, buf, buf, None)
    assert rc == -1 and b"B <= 32" in so.lab4d_last_error()
    rc = so.lab4d_skel_bones_forward(buf, None, buf, buf, buf, buf, buf, buf, 1, 25, buf, buf, None)
    assert rc == -1 and b"null" in so.lab4d_last_error()
    assert so.lab4d_fk_forward(buf, buf, None, buf, buf, 0, 25, 0, buf, buf, None) == 0  # empty input: nothing to launch


def test_pose_refuses_cpu_tensors():
    import torch
    from lab4d_amd import pose
    with pytest.raises(RuntimeError, match="no CPU path"):
        pose.fk_se3(torch.zeros(2, 3, 3), torch.zeros(2, 3, 3), {1: 0, 2: 1, 3: 2})


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
    import ast
    pkg = os.path.join(ROOT, "lab4d_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                if any(n == "oracle" or n.startswith("oracle.") for n in names):
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    # bench.py: the oracle is imported inside cpu_baseline() only
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert uses == (fn.name == "cpu_baseline"), fn.name
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n) for n in tree.body)


def test_rowmlp_programs_are_validated_before_any_launch(so):
    """include/lab4d_rowmlp.h's contract on the CPU (argument checks run before the launch): a layer writing into its own input, two layers writing
    overlapping columns, columns leaving the strip, too many layers -- all refused with LAB4D_EINVAL and a message."""
    from lab4d_amd import rowmlp
    so.lab4d_last_error.restype = ctypes.c_char_p
    so.lab4d_rowmlp_forward.argtypes = [ctypes.POINTER(rowmlp._Prog), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]

    def prog(layers, stride=64):
        p = rowmlp._Prog()
        p.n_layers, p.row_stride = len(layers), stride
        for i, (i_dim, o_dim, src, dst) in enumerate(layers):
            q = p.layer[i]
            q.W, q.in_dim, q.out_dim, q.src_col, q.dst_col = 0x1000, i_dim, o_dim, src, dst  # (never dereferenced: the checks fail first)
        return p

    fake = ctypes.c_void_p(0x2000)
    for layers, word in [([(8, 8, 0, 4)], b"own input"), ([(8, 8, 0, 16), (8, 8, 16, 20)], b"own input"), ([(8, 8, 0, 16), (8, 8, 0, 20)], b"overlapping"),
                         ([(8, 8, 0, 60)], b"leave the row strip"), ([(2000, 8, 0, 16)], b"outside")]:
        p = prog(layers, stride=64 if layers[0][0] < 100 else 4096)
        assert so.lab4d_rowmlp_forward(ctypes.byref(p), fake, 4, None) == -1
        assert word in so.lab4d_last_error(), (layers, so.lab4d_last_error())
    p = prog([(8, 8, 0, 16)])
    p.n_layers = 17
    assert so.lab4d_rowmlp_forward(ctypes.byref(p), fake, 4, None) == -1


def test_rowmlp_ctypes_structs_have_the_c_layout(tmp_path):
    """lab4d_amd/rowmlp.py mirrors lab4d_rowmlp_prog / _layer / _io with ctypes.Structure: sizes and the offsets of the arrays must equal what a C compiler
    gives include/lab4d_rowmlp.h (a field added on one side only would shift every pointer behind it)."""
    import subprocess
    from lab4d_amd import rowmlp
    src = tmp_path / "sz.c"
    src.write_text('#include <stdint.h>\n#include <stddef.h>\n#include <stdio.h>\n#include "lab4d_rowmlp.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(lab4d_rowmlp_prog), sizeof(lab4d_rowmlp_layer), sizeof(lab4d_rowmlp_io),'
                   ' offsetof(lab4d_rowmlp_prog, in), offsetof(lab4d_rowmlp_prog, out), offsetof(lab4d_rowmlp_prog, layer), offsetof(lab4d_rowmlp_prog, max_ts));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    c = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    P = rowmlp._Prog
    assert c == [ctypes.sizeof(P), ctypes.sizeof(rowmlp._Layer), ctypes.sizeof(rowmlp._IO), P.inp.offset, P.out.offset, P.layer.offset, P.max_ts.offset]
