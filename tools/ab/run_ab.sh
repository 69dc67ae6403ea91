# A/B of two builds of the library on ONE box, interleaved: the in-tree build (new) against one built with tools/ab/gemm_common_prev.h
# (produce that file with `git show <rev>:eoe_amd/csrc/gemm_common.h > tools/ab/gemm_common_prev.h`; the GPU box has no git history)
set -e
cd $GRAFT_REPO_ROOT
cp eoe_amd/libeoe_hip.so /tmp/new.so; cp eoe_amd/libeoe_hip.so.stamp /tmp/new.stamp
cp eoe_amd/csrc/gemm_common.h /tmp/new_common.h
cp tools/ab/gemm_common_prev.h eoe_amd/csrc/gemm_common.h
python -c "from eoe_amd import _build; _build.build(force=True, verbose=False)" > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
cp eoe_amd/libeoe_hip.so /tmp/old.so; cp eoe_amd/libeoe_hip.so.stamp /tmp/old.stamp
cp eoe_amd/csrc/gemm_common.h /tmp/old_common.h
# (the loader checks the library's content stamp against the sources: each run sees the header its library was built from)
run() { cp /tmp/$1.so eoe_amd/libeoe_hip.so; cp /tmp/$1.stamp eoe_amd/libeoe_hip.so.stamp; cp /tmp/$1_common.h eoe_amd/csrc/gemm_common.h; EOE_PROF_SHAPES=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline 2>/dev/null > gpurun_out/sh/ab_$1_$2.json; }
mkdir -p gpurun_out/sh
for r in 1 2 3; do run old $r; run new $r; done
cp /tmp/new.so eoe_amd/libeoe_hip.so; cp /tmp/new.stamp eoe_amd/libeoe_hip.so.stamp; cp /tmp/new_common.h eoe_amd/csrc/gemm_common.h
python - <<'PY'
import json
res, steps = {}, {}
for v in ("old", "new"):
    for r in (1, 2, 3):
        d = json.loads(open(f"gpurun_out/sh/ab_{v}_{r}.json").read())
        steps.setdefault(v, []).append(d["ms_per_step"])
        for n, x in d["roofline"]["kernels_ms_per_step"].items():
            if n.startswith("nt_128"):
                res.setdefault(n, {}).setdefault(v, []).append(x)
print("step ms:", steps)
for n in sorted(res, key=lambda n: -sum(res[n]["old"])):
    a, b = res[n]["old"], res[n]["new"]
    print(f"{n:30s} old {sum(a)/len(a):7.3f}  new {sum(b)/len(b):7.3f}  diff {sum(b)/len(b)-sum(a)/len(a):+.3f}")
PY
