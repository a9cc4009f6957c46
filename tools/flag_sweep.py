"""Development aid: the library built several times with different code-generation flags for the detection kernels
(csrc/stream_kernels.hip only; every other object is shared), one .so per variant under rtl_433_amd/lib/variants/.
    python tools/flag_sweep.py build            # cross-compiles here, prints registers / scratch of the bench's k_wave form
    python tools/flag_sweep.py run [tags...]    # on the GPU box: tools/kbench.py --lib per variant + a digest of the records
The variants are git-ignored like every built file and travel to the GPU box with the snapshot."""
import hashlib, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rtl_433_amd import build as B

VAR_DIR = os.path.join(B.OUT_DIR, "variants")
TMP = os.environ.get("FS_TMP", "/tmp/fs")
VARIANTS = {
    "base": [],
    "maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "bias0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "ifcvt": ["-mllvm", "-amdgpu-early-ifcvt"],
    "o2": ["-O2"],
    "noalign": ["-mllvm", "-amdgpu-disable-loop-alignment"],
    "wprio": ["-mllvm", "-amdgpu-set-wave-priority"],
}
KERNEL = "k_wave<2, true, true, false, 2>"


def compile_one(src, extra, obj, remarks=False):
    cmd = [B._hipcc()] + B.FLAGS + extra + ["-I", B.INC, "-I", B.CSRC, "-c", os.path.join(B.CSRC, src), "-o", obj]
    if remarks:
        cmd += ["-Rpass-analysis=kernel-resource-usage"]
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def resources(text):
    """registers / scratch / occupancy of every k_wave instantiation from the compiler's remarks"""
    out, name = {}, None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = name.replace("r433::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            out[name] = {}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                out[name][key] = int(m.group(1))
    return out


def build(tags):
    os.makedirs(VAR_DIR, exist_ok=True)
    common = os.path.join(TMP, "common")
    os.makedirs(common, exist_ok=True)
    procs = []
    for src in B.SOURCES:
        if src == "stream_kernels.hip":
            continue
        obj = os.path.join(common, src.rsplit(".", 1)[0] + ".o")
        if not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(os.path.join(B.CSRC, src)):
            procs.append((src, compile_one(src, [], obj)))
    for tag in tags:
        procs.append((tag, compile_one("stream_kernels.hip", VARIANTS[tag], os.path.join(TMP, f"sk_{tag}.o"), remarks=True)))
    for what, p in procs:
        out, _ = p.communicate()
        text = out.decode(errors="replace")
        if p.returncode != 0:
            print(f"{what}: hipcc FAILED\n{text[-1500:]}")
            continue
        if what in VARIANTS:
            res = resources(text)
            print(f"{what:10s} {KERNEL}: {res.get(KERNEL)}", flush=True)
            with open(os.path.join(TMP, f"sk_{what}.res"), "w") as f:
                for k, v in sorted(res.items()):
                    f.write(f"{k} {v}\n")
            objs = [os.path.join(common, s.rsplit(".", 1)[0] + ".o") for s in B.SOURCES if s != "stream_kernels.hip"] + [os.path.join(TMP, f"sk_{what}.o")]
            so = os.path.join(VAR_DIR, f"librtl433hip_{what}.so")
            subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC",
                                   "-Wl,--version-script=" + os.path.join(B.CSRC, "exports.map"), "-o", so] + objs)


def run(tags, streams=8192, reps=6):
    """Every variant in a process of its own (tools/kbench.py --lib), the same tiled input; prints one line per variant."""
    import numpy as np
    for tag in tags:
        so = os.path.join(VAR_DIR, f"librtl433hip_{tag}.so")
        if not os.path.exists(so):
            print(f"{tag}: not built")
            continue
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variant_bench.py"), so, str(streams), str(reps)],
                           capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l]
        print(f"{tag:10s} " + (lines[-1] if lines else f"rc={r.returncode}"), flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "build"
    tags = sys.argv[2:] or list(VARIANTS)
    build(tags) if cmd == "build" else run(tags)
