"""Build recipe for libmodest_hip.so (hipcc, gfx950 only, in-tree).

``python -m modest_amd.build`` compiles every ``csrc/*.hip`` translation unit
for gfx950 and links them into ``modest_amd/lib/libmodest_hip.so``.  hipcc
cross-compiles without a GPU, so this also runs in the GPU-less build
container.  The library is kept in-tree (git-ignored) so that it travels with
the repository snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
OBJDIR = LIBDIR / "obj"
LIB = LIBDIR / "libmodest_hip.so"

ARCH = "gfx950"
# -ffp-contract=off: every kernel states its fused multiply-adds explicitly
# (fma()/fmaf()); parity with the reference arithmetic depends on it.
CXXFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fno-fast-math", "-Wall", "-Wno-unused-function", f"-I{ROOT / 'include'}", f"-I{CSRC}",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libmodest_hip.so")
    return exe


def _digest(paths) -> str:
    h = hashlib.sha256()
    h.update(" ".join(CXXFLAGS).encode())
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def sources():
    return sorted(CSRC.glob("*.hip"))


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs = sources()
    hdrs = sorted(CSRC.glob("*.h")) + sorted((ROOT / "include").glob("*.h"))
    stamp = LIBDIR / "build.stamp"
    digest = _digest(srcs + hdrs)
    if (not force and LIB.exists() and (LIBDIR / "kernel_resources.json").exists() and stamp.exists()
            and stamp.read_text() == digest):
        return LIB
    OBJDIR.mkdir(parents=True, exist_ok=True)
    for stale in set(OBJDIR.glob("*.o")) - {OBJDIR / (x.stem + ".o") for x in srcs}:
        stale.unlink()   # object of a source file that no longer exists
    hipcc = _hipcc()

    resources = {}

    def compile_one(src: Path) -> Path:
        obj = OBJDIR / (src.stem + ".o")
        # the resource remarks (registers, scratch, LDS per kernel) are kept next to the library:
        # a kernel that uses scratch memory pays ~25 us at dispatch, tests/ assert there is none
        cmd = [hipcc, *CXXFLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[modest_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
        rest = []
        cur = None
        for line in r.stderr.splitlines():
            if "[-Rpass-analysis=kernel-resource-usage]" not in line:
                if not rest or rest[-1] != "skip":
                    rest.append(line)
                continue
            body = line.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
            key, _, val = body.partition(":")
            if key == "Function Name":
                cur = resources.setdefault(val.strip(), {"file": src.name})
            elif cur is not None and val.strip().lstrip("-").isdigit():
                cur[key.strip()] = int(val.strip())
        # drop the source-context lines clang prints under every remark
        diag = [l for l in rest if l.strip() and not l.lstrip().startswith(("|", "^", "In file included"))
                and not l.lstrip()[:1].isdigit()]
        if verbose and diag:
            print("\n".join(diag), file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    (LIBDIR / "kernel_resources.json").write_text(json.dumps(resources, indent=1, sort_keys=True))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print("[modest_amd.build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
