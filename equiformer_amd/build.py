"""Build libequiformer_hip.so (all HIP kernels + the C ABI of include/equiformer_hip.h) for gfx950.

    python -m equiformer_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is built IN-TREE (equiformer_amd/libequiformer_hip.so) so that it
travels with the repository snapshot to the GPU box; it links against libamdhip64 only (no torch).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libequiformer_hip.so")
SOURCES = ["gemm.hip", "gemmx.hip", "sfc.hip", "sfcx.hip", "sfcy.hip", "sfcw.hip", "rowops.hip", "edge.hip", "graph.hip", "second.hip", "dpattn.hip", "optim.hip", "prof.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
# per-source flags.  sfcx.hip issues bf16 MFMAs: no packed-FP32 VALU instruction may be generated beside them (the SLP
# vectoriser is what turns neighbouring scalar fp32 operations into v_pk_*_f32)
EXTRA_FLAGS = {"sfcx.hip": ["-fno-slp-vectorize"], "sfcy.hip": ["-fno-slp-vectorize"], "sfcw.hip": ["-fno-slp-vectorize"], "gemmx.hip": ["-fno-slp-vectorize"]}


def _code_only(text):
    """C / C++ source without comments, runs of whitespace collapsed to one space (string / character literals kept verbatim;
    a ' that follows an alphanumeric character is a C++14 digit separator, not a literal)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or (c == "'" and not (i > 0 and (text[i - 1].isalnum() or text[i - 1] == "_"))):
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            if out and out[-1] != " ":
                out.append(" ")
        elif c.isspace():
            if out and out[-1] != " ":
                out.append(" ")
            i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out).strip()


def source_hash():
    """sha256 over the CODE of the HIP sources + the public header (comments and whitespace stripped: a build's identity
    does not change with its documentation): identifies the build a measurement (profiles/pmc_dominant.json) belongs to."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(HERE, "..", "include", "equiformer_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(_code_only(open(f, "r", encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "equiformer_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out=None, extra_flags=(), only=None):
    """out / extra_flags: an A/B VARIANT of the library (development: `python -m equiformer_amd.build --variant NAME -DEQF_X=0`
    writes equiformer_amd/libequiformer_hip_NAME.so, objects under csrc/.variant_NAME/; loaded with EQF_LIB_VARIANT=NAME)."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    # development builds: EQF_EXTRA_FLAGS="-DEQF_DEV_SWITCHES=1" (phase switches / cycle counters inside the sfc and gemm
    # kernels, tools/sfc_exp.py, tools/gemm_exp.py) or "-DEQF_XTRACE=1" (in-kernel clock samples, tools/sfcx_trace.py)
    dev = os.environ.get("EQF_EXTRA_FLAGS", "").split() + list(extra_flags)
    objs = []
    procs = []
    objdir = CSRC
    if out is not None:
        objdir = os.path.join(CSRC, ".variant_" + os.path.basename(out).replace("libequiformer_hip_", "").replace(".so", ""))
        os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "equiformer_hip.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    incremental = out is None and not dev and os.environ.get("EQF_BUILD_ALL", "") != "1"
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        extra = list(EXTRA_FLAGS.get(s, []))
        # the flags an object was compiled with are recorded beside it: an object left behind by a development build
        # (EQF_EXTRA_FLAGS) or by an older FLAGS / EXTRA_FLAGS is rebuilt, not linked into the product library
        flag_file = o + ".flags"
        flag_text = " ".join(FLAGS + extra + dev)
        same_flags = os.path.exists(flag_file) and open(flag_file).read() == flag_text
        if (incremental and s != "rowops.hip" and os.path.exists(o) and same_flags
                and os.path.getmtime(o) > max(os.path.getmtime(os.path.join(CSRC, s)), newest_header)):
            continue  # object newer than its source and every header (rowops.hip always: it carries the source hash)
        if only is not None and s not in only:
            base = os.path.join(CSRC, s.replace(".hip", ".o"))
            if os.path.exists(base):  # (variant builds: the product build's objects of untouched sources are reused)
                objs[-1] = base
                continue
        if s == "rowops.hip":  # eqf_version() lives there and reports the hash of the sources of THIS build
            extra.append('-DEQF_SOURCE_HASH="%s"' % source_hash())
        cmd = [hipcc] + FLAGS + extra + dev + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(flag_file):
            os.remove(flag_file)
        procs.append((cmd, subprocess.Popen(cmd), flag_file, flag_text))
    for cmd, p, flag_file, flag_text in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        with open(flag_file, "w") as fh:
            fh.write(flag_text)
    target = out or LIB
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        name = sys.argv[i + 1]
        rest = [a for a in sys.argv[i + 2:] if a != "--force"]
        only = [a[len("--only="):] for a in rest if a.startswith("--only=")]
        flags = [a for a in rest if not a.startswith("--only=")]
        print(build(force=True, out=os.path.join(HERE, "libequiformer_hip_%s.so" % name), extra_flags=flags,
                    only=only[0].split(",") if only else None))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
