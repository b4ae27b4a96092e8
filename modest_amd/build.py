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
] + os.environ.get("MODEST_EXTRA_CXXFLAGS", "").split()   # (experiments: -DB4_JT_=640 ...; part of the build digest)


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


# ---- the ordering the last-block reductions rest on, checked in the machine code ----------------------------------
# plane.hip / boxfit.hip publish per-block partial results with agent-scope (sc1, write-through) stores and take a ticket;
# the block that draws the last ticket reads the partials.  What orders the stores before the ticket is NOT the HIP memory
# model but `modest_drain_stores()` (common.h: an inline `s_waitcnt vmcnt(0)`) ahead of the workgroup barrier that
# precedes the ticket -- rounds 1-2 shipped without it (0.08 % wrong planes / boxes under load).  A compiler upgrade may
# move or drop that wait, so the build disassembles those translation units and refuses to produce a library in which,
# in any kernel, the last sc1 store ahead of a returning atomic add (the ticket) is not followed by `s_waitcnt vmcnt(0)`
# before the ticket -- and before the workgroup barrier, where one lies between them (the ticket is then taken by
# another wavefront than the one that stored).
DRAIN_SOURCES = ("plane.hip", "boxfit.hip")


def ticket_drain_violations(asm: str) -> list:
    """[(kernel, line number of the ticket atomic, reason)] for device assembly text (hipcc -S --cuda-device-only)."""
    import re
    bad, kernel, body, start = [], None, [], 0
    lines = asm.splitlines()

    def check(name, rows, first):
        for t, row in enumerate(rows):
            if not (re.search(r"\b(global|flat)_atomic_add(_u32)?\b", row) and re.search(r"\bsc0\b", row)):
                continue   # a returning 32-bit atomic add = a ticket
            stores = [i for i in range(t) if re.search(r"\b(global|flat)_store_\w+", rows[i]) and re.search(r"\bsc1\b", rows[i])]
            if not stores:
                continue   # a ticket without published partials (compaction cursors): nothing to order
            s = stores[-1]
            wait = next((i for i in range(s + 1, t) if re.search(r"\bs_waitcnt\b.*\bvmcnt\(0\)", rows[i])), None)
            barrier = next((i for i in range(s + 1, t) if re.search(r"\bs_barrier\b", rows[i])), None)
            if wait is None:
                bad.append((name, first + t + 1, "no s_waitcnt vmcnt(0) between the last sc1 store and the ticket"))
            elif barrier is not None and wait > barrier:
                bad.append((name, first + t + 1, "the s_waitcnt vmcnt(0) comes after the barrier that precedes the ticket: another "
                                                  "wavefront's store may still be in flight"))

    for n, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if kernel:
                check(kernel, body, start)
            kernel, body, start = m.group(1), [], n + 1
        elif kernel is not None:
            body.append(line)
            if re.search(r"\bs_endpgm\b", line):
                check(kernel, body, start)
                kernel, body = None, []
    return bad


def device_asm(src: Path, out: Path) -> str:
    """gfx950 assembly of one translation unit (seconds)."""
    cmd = [_hipcc(), *[f for f in CXXFLAGS if f not in ("-fPIC",)], "-S", "--cuda-device-only", str(src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed on {src.name}:\n{r.stderr}")
    return out.read_text()


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

    def drain_check(src: Path):
        bad = ticket_drain_violations(device_asm(src, OBJDIR / (src.stem + ".gfx950.s")))
        if bad:
            raise RuntimeError(f"{src.name}: a last-block reduction publishes its partial results without draining them before "
                               f"the ticket (common.h: modest_drain_stores): {bad}")

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        checks = [ex.submit(drain_check, x) for x in srcs if x.name in DRAIN_SOURCES]
        objs = list(ex.map(compile_one, srcs))
        for c in checks:
            c.result()
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
