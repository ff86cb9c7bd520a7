#!/usr/bin/env python3
"""Compile-and-link check of INTEGRATION.md Level 0a (round 6, review item 7): the reference's OWN pybind glue
(`GAN-Based-SR/basicsr/losses/similarity/similaritywrapper.cpp`, which calls `_compute_similarity` /
`_compute_similarity_backward` of `similarity.h:2-23`) is built by `torch.utils.cpp_extension.load` exactly as
`similaritywrapper.py:15-23` builds it -- minus `similarity.cu`, plus `-lssg_hip` -- and imported.  An unresolved or
differently mangled symbol fails the link or the import; no GPU is needed (nothing is launched).

Runs in the BUILD CONTAINER only: it needs /root/reference (which never travels to the GPU box).  The reference file is
read where it lies; on ROCm `load()` hipifies its sources in place, so the one .cpp is staged in a temporary directory
that is deleted afterwards -- no reference text enters the repo, nothing is written under /root/reference.

    python tools/level0a_link_check.py        ->  profiles/r6_level0a_link_check.txt
"""
import os
import shutil
import subprocess
import sys
import tempfile

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/GAN-Based-SR/basicsr/losses/similarity"
CSRC = os.path.join(ROOT, "ssl_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "r6_level0a_link_check.txt")


def main():
    if not os.path.isdir(REF):
        print("no /root/reference here: Level 0a can only be checked in the build container")
        return 2
    if not os.path.exists(os.path.join(CSRC, "libssg_hip.so")):
        subprocess.check_call(["make", "-C", CSRC, "-j4", "libssg_hip.so"])
    import torch
    from torch.utils.cpp_extension import load
    tmp = tempfile.mkdtemp(prefix="level0a_")
    log = []
    try:
        src = os.path.join(tmp, "similaritywrapper.cpp")
        shutil.copyfile(os.path.join(REF, "similaritywrapper.cpp"), src)      # ephemeral staging for load()'s hipify pass
        # the header the glue includes: the repo's include/similarity.h (declares the reference's two functions with the
        # reference's parameter lists; compared against the reference's own header below)
        mod = load(name="compute_similarity", sources=[src], with_cuda=True, build_directory=tmp, verbose=False,
                   extra_include_paths=[os.path.join(ROOT, "include")],
                   extra_ldflags=["-L" + CSRC, "-lssg_hip", "-Wl,-rpath," + CSRC])
        have = [n for n in ("compute_similarity", "compute_similarity_backward") if callable(getattr(mod, n, None))]
        so = [f for f in os.listdir(tmp) if f.endswith(".so")]
        log.append("torch %s, load(name='compute_similarity', sources=[similaritywrapper.cpp], -lssg_hip): built %s" % (torch.__version__, so))
        log.append("module exports: %s" % have)
        nm = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(tmp, so[0])], stdout=subprocess.PIPE, text=True).stdout
        und = [l.split()[-1] for l in nm.splitlines() if "_compute_similarity" in l]
        log.append("undefined symbols the glue takes from libssg_hip.so: %s" % und)
        exp = subprocess.run(["nm", "-D", "--defined-only", os.path.join(CSRC, "libssg_hip.so")], stdout=subprocess.PIPE, text=True).stdout
        ok = len(have) == 2 and len(und) == 2 and all(any(u == l.split()[-1] for l in exp.splitlines()) for u in und)
        # the declarations: the reference's header against include/similarity.h, parameter list by parameter list
        import re
        def protos(path):
            txt = re.sub(r"//[^\n]*|/\*.*?\*/", " ", open(path).read(), flags=re.S)
            return sorted(re.sub(r"\s+", " ", m).strip() for m in re.findall(r"void\s+_compute_similarity\w*\s*\([^)]*\)", txt))
        a, b = protos(os.path.join(REF, "similarity.h")), protos(os.path.join(ROOT, "include", "similarity.h"))
        norm = lambda ps: [re.sub(r"\b(const\s+)?(float|int)\s*\*?\s*\w+\s*(?=[,)])", lambda m: m.group(0).rsplit(" ", 1)[0].replace(" ", "") + " ", p) for p in ps]
        same = [re.sub(r"\s+", "", x) for x in norm(a)] == [re.sub(r"\s+", "", x) for x in norm(b)]
        log.append("declarations (names dropped) identical to the reference's similarity.h: %s" % same)
        ok = ok and same
        log.append("LEVEL 0a LINK CHECK: %s" % ("PASS" if ok else "FAIL"))
        rc = 0 if ok else 1
    except Exception as e:   # noqa: BLE001 -- the log is the product
        log.append("LEVEL 0a LINK CHECK: FAIL (%s: %s)" % (type(e).__name__, str(e)[-2000:]))
        rc = 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    text = "\n".join(log) + "\n"
    print(text, end="")
    with open(OUT, "w") as f:
        f.write("# python tools/level0a_link_check.py (build container, no GPU)\n" + text)
    return rc


if __name__ == "__main__":
    sys.exit(main())
