"""Build the HIP library (libozk_hip.so) and the JNI shim libraries in-tree for gfx950.

    python -m octopuszk_amd.build [--force]

hipcc cross-compiles without a GPU.  Outputs land next to this file so they travel
with the repo snapshot to the GPU box (they are git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libozk_hip.so")

HIP_SOURCES = ["msm_var.hip", "msm_var_g2.hip", "msm_fixed.hip", "fft.hip", "host_ctx.hip"]
HEADERS = ["consts_gen.h", "mad_chain_gen.h", "fp29.cuh", "fq2.cuh", "ec.cuh", "quad.cuh", "curve.cuh", "msm_var.cuh", "msm_var_driver.cuh", "glv.cuh", "ozk_common.h", "host_ctx.h", "pin_cache.h",
           os.path.join("..", "..", "include", "ozk.h")]
JNI_LIBS = {
    "libAlgebraMSMVariableBaseMSM.so": "jni_var_msm.c",
    "libAlgebraMSMFixedBaseMSM.so": "jni_fixed_msm.c",
    "libAlgebraFFTAuxiliary.so": "jni_fft.c",
}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU path)")


def build(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    if force or _newer(LIB, deps):
        defs = []
        if os.path.exists(os.path.join(CSRC, "fq2.cuh")):
            defs.append("-DOZK_WITH_G2")
        # one object per translation unit, compiled in parallel, then one device link
        objdir = os.path.join(HERE, "_obj")
        os.makedirs(objdir, exist_ok=True)
        common = [hipcc(), "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc",
                  "-Wno-unused-result", "-Wno-pass-failed"] + defs
        procs, objs = [], []
        for src in srcs:
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            objs.append(obj)
            if force or _newer(obj, [src] + [os.path.join(CSRC, h) for h in HEADERS]):
                cmd = common + ["-c", "-o", obj, src]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        for cmd, pr in procs:
            out, _ = pr.communicate()
            if pr.returncode != 0:
                sys.stderr.write(out.decode(errors="replace"))
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # JNI shims: plain C++ (g++), forward to libozk_hip.so through the C ABI
    for lib, src in JNI_LIBS.items():
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        out = os.path.join(HERE, lib)
        if force or _newer(out, [s, os.path.join(CSRC, "jni_common.h"), os.path.join(ROOT, "include", "ozk_jni.h"),
                                 os.path.join(ROOT, "include", "ozk.h")]):
            cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                   "-o", out, s, "-L", HERE, "-lozk_hip", "-Wl,-rpath,$ORIGIN"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
