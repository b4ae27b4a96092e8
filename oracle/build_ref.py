"""ORACLE BUILD RECIPE — test infrastructure only.

Compiles the reference's own ``iou3d_cpu.cpp`` (the CPU twin of the CUDA BEV-IoU
kernel, same arithmetic line for line: src/iou3d_cpu.cpp:128-229 vs
src/iou3d_nms_kernel.cu:104-234) FROM /root/reference, where it lies, into
``oracle/_ref/iou3d_ref.so`` (git-ignored).  Headers it needs beyond torch:
``cuda.h`` / ``cuda_runtime_api.h`` — the genuine CUDA runtime headers that this
image ships inside triton's nvidia backend are used; no stand-in is written.
Only the build container has /root/reference; the GPU box uses the prebuilt
file.  The CUDA sources (iou3d_nms_kernel.cu, iou3d_nms.cpp) are unbuildable
here (no nvcc / CUDA runtime library) and are not attempted.
"""
import os
import subprocess
import sys
import sysconfig
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF_SRC = Path("/root/reference/generate_cluster_mask/utils/iou3d_nms/src")
OUT = HERE / "_ref" / "iou3d_ref.so"


def build(force: bool = False):
    if not REF_SRC.exists():
        return OUT if OUT.exists() else None
    if OUT.exists() and not force:
        return OUT
    import torch
    from torch.utils import cpp_extension
    import triton
    cuda_inc = Path(triton.__file__).parent / "backends" / "nvidia" / "include"
    if not (cuda_inc / "cuda_runtime_api.h").exists():
        print("CUDA runtime headers not present in this image: reference IoU unbuildable", file=sys.stderr)
        return None
    OUT.parent.mkdir(parents=True, exist_ok=True)
    incs = cpp_extension.include_paths() + [sysconfig.get_paths()["include"], str(REF_SRC), str(cuda_inc)]
    tlib = Path(torch.__file__).parent / "lib"
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w",
           "-DTORCH_EXTENSION_NAME=iou3d_ref", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in incs]
    cmd += [str(REF_SRC / "iou3d_cpu.cpp"), str(HERE / "ref_iou_bind.cpp"), "-o", str(OUT),
            f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", f"-Wl,-rpath,{tlib}"]
    print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
