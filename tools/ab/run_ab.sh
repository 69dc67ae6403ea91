# A/B of two builds of the library on ONE box, interleaved: the in-tree build (new) against one built with the files under tools/ab/prev/
# swapped in (same relative paths; produce them with `git show <rev>:<path> > tools/ab/prev/<path>` -- the GPU box has no git history).
# Usage: bash tools/ab/run_ab.sh [bench.py arguments]; EOE_PROF_SHAPES=1 for per-shape NT GEMM times.
set -e
cd $GRAFT_REPO_ROOT
ARGS="${@:---steps 20 --warmup 5 --no-cpu-baseline --no-torch-baseline --no-roofline}"
FILES=$(cd tools/ab/prev && find . -type f | sed 's|^\./||')
mkdir -p /tmp/ab/new /tmp/ab/old gpurun_out/sh
for f in $FILES; do mkdir -p /tmp/ab/new/$(dirname $f); cp $f /tmp/ab/new/$f; done
cp eoe_amd/libeoe_hip.so /tmp/ab/new.so; cp eoe_amd/libeoe_hip.so.stamp /tmp/ab/new.stamp
for f in $FILES; do cp tools/ab/prev/$f $f; done
python -c "from eoe_amd import _build; _build.build(force=True, verbose=False)" > /tmp/ab/build.log 2>&1 || { tail -5 /tmp/ab/build.log; exit 1; }
cp eoe_amd/libeoe_hip.so /tmp/ab/old.so; cp eoe_amd/libeoe_hip.so.stamp /tmp/ab/old.stamp
# (the loader checks the library's content stamp against the sources: each run sees the sources its library was built from)
run() {
    cp /tmp/ab/$1.so eoe_amd/libeoe_hip.so; cp /tmp/ab/$1.stamp eoe_amd/libeoe_hip.so.stamp
    for f in $FILES; do if [ $1 = old ]; then cp tools/ab/prev/$f $f; else cp /tmp/ab/new/$f $f; fi; done
    python bench.py $ARGS 2>/dev/null > gpurun_out/sh/ab_$1_$2.json
}
for r in 1 2 3; do run old $r; run new $r; done
run new 0
python - <<'PY'
import json
res, steps = {}, {}
for v in ("old", "new"):
    for r in (1, 2, 3):
        d = json.loads(open(f"gpurun_out/sh/ab_{v}_{r}.json").read())
        steps.setdefault(v, []).append(d["ms_per_step"])
        for n, x in (d.get("roofline") or {}).get("kernels_ms_per_step", {}).items():
            res.setdefault(n, {}).setdefault(v, []).append(x)
print("step ms:", steps)
for n in sorted(res, key=lambda n: -sum(res[n]["old"])):
    a, b = res[n]["old"], res[n]["new"]
    print(f"{n:30s} old {sum(a)/len(a):7.3f}  new {sum(b)/len(b):7.3f}  diff {sum(b)/len(b)-sum(a)/len(a):+.3f}")
PY
