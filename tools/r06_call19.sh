#!/bin/bash
# Round 6, call 19: are the shipped GEMM plans (tuned in round 3-5) still the best on this build?  Re-tune fp32 B=10 H=256 from scratch into scratch, A/B the step
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
cp transfuser_amd/plans/mi355x.txt $O/plans_shipped.txt
TF_RETUNE=1 timeout 600 python tools/tune.py $O/plans_retuned_r06.txt 10 256 fp32 2>&1 | grep -v Warn | tail -3
python - <<PY
a=[l for l in open("$O/plans_shipped.txt") if not l.startswith("#")]
b=[l for l in open("$O/plans_retuned_r06.txt") if not l.startswith("#")]
ka={";".join(l.split(";")[:6]):l for l in a}; kb={";".join(l.split(";")[:6]):l for l in b}
diff=[k for k in kb if k in ka and ka[k]!=kb[k]]
print("plans: shipped %d, retuned %d, common keys with a different choice: %d" % (len(a), len(b), len(diff)))
for k in diff[:25]: print("  ", ka[k].strip(), "->", kb[k].strip().split(";",6)[-1])
PY
for rep in 1 2 3; do
  timeout 200 $B 2>/dev/null | bl "fp32 shipped plans "
  TF_PLANS=$O/plans_retuned_r06.txt timeout 200 $B 2>/dev/null | bl "fp32 re-tuned plans"
done
