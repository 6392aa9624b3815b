"""north_star: "the Cython host in align.pyx becomes a thin C-ABI shim".  INTEGRATION.md §2-3 shows the declarations and the call
sites; here they are COMPILED (Cython 3 is in the image): pywfa_amd/cython_shim/wfa_hip.pxd + wfa_shim.pyx are cythonized, built
against include/wfa_hip.h and linked to libwfa_hip.so in a temporary directory, and the module is exercised — on CPU the host-only
entry points (ABI version, default configuration, validation, the pretty printer), on the GPU box one alignment per call with
pywfa's known answer (pywfa/README.rst:32-43)."""
import importlib.util
import os
import subprocess
import sys
import sysconfig

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SHIM = os.path.join(ROOT, "pywfa_amd", "cython_shim")


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    build = tmp_path_factory.mktemp("cython_shim")
    for f in ("wfa_hip.pxd", "wfa_shim.pyx"):
        (build / f).write_text(open(os.path.join(SHIM, f)).read())
    subprocess.run([sys.executable, "-m", "cython", "-3", "wfa_shim.pyx"], cwd=build, check=True)
    so = build / ("wfa_shim" + sysconfig.get_config_var("EXT_SUFFIX"))
    libdir = os.path.join(ROOT, "pywfa_amd")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wno-deprecated-declarations", "-I", os.path.join(ROOT, "include"),
                    "-I", sysconfig.get_paths()["include"], "wfa_shim.c", "-o", str(so),
                    "-L", libdir, "-lwfa_hip", f"-Wl,-rpath,{libdir}"], cwd=build, check=True)
    spec = importlib.util.spec_from_file_location("wfa_shim", str(so))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shim_compiles_and_binds_the_header(shim):
    from pywfa_amd import _native
    assert shim.abi_version() == _native.ABI_VERSION
    a = shim.ShimAligner("ACGT", create=False)      # configuration only: no device needed
    c = a.config
    assert (c["distance"], c["match"], c["mismatch"], c["gap_opening"], c["gap_extension"]) == (3, 0, 4, 6, 2)
    assert (c["scope"], c["span"], c["xdrop"], c["min_wavefront_length"], c["wildcard"]) == (1, 1, 20, 10, -1)   # align.pyx:309-334
    with pytest.raises(ValueError):
        shim.ShimAligner("ACGT", mismatch=0, create=False)   # the reference exit(1)s (wavefront_penalties.c:101-112)
    with pytest.raises(RuntimeError):
        a.wavefront_align("ACGT")


def test_shim_pretty_printer(shim):
    from pywfa_amd import _native
    ops, p, t = b"MMXMMIMM", b"ACGTACG", b"ACCTAGCG"
    assert shim.sprint_pretty(ops, p, t) == _native.cigar_sprint_pretty(bytearray(ops), p, t)


@pytest.mark.gpu
def test_shim_aligns_one_pair_per_call(gpu, shim):
    a = shim.ShimAligner("TCTTTACTCGCGCGTTGGAGAAATACAATAGT")
    assert a.wavefront_align("TCTATACTGCGCGTTTGGAGAAATAAAATAGT") == -24      # pywfa/README.rst:32-43
    assert a.status == 0 and a.cigarstring == "3M1X4M1D7M1I9M1X6M"
    assert a.wavefront_align("ACGTTAGC", pattern="ACGTAGC") == -8 and a.cigarstring == "4M1I3M"
    b = shim.ShimAligner("ACGTAGC", scope="score", span="end-to-end", mismatch=5)
    assert b.wavefront_align("ACGTTAGC") == -8 and b.cigarstring == ""
