"""Build pywfa_amd/host/_host<ext-suffix>.so in-tree: cython -> C, gcc -fopenmp (called by __graft_entry__.build())."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    so = os.path.join(HERE, "_host" + sysconfig.get_config_var("EXT_SUFFIX"))
    srcs = [os.path.join(HERE, f) for f in ("_host.pyx", "host_core.c")]
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    import numpy
    subprocess.run([sys.executable, "-m", "cython", "-3", "_host.pyx", "-o", "_host.c"], cwd=HERE, check=True)
    subprocess.run(["gcc", "-shared", "-fPIC", "-O3", "-fopenmp", "-Wno-deprecated-declarations", "-Wno-unused-function",
                    "-I", sysconfig.get_paths()["include"], "-I", numpy.get_include(), "-I", HERE, "_host.c", "-o", so], cwd=HERE, check=True)
    os.remove(os.path.join(HERE, "_host.c"))
    return so


if __name__ == "__main__":
    print(build(force=True))
