mkdir -p gpurun_out
run() {
  tag=$1; n=$2; shift; shift
  env "$@" timeout 600 python bench.py --n $n --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g30_$tag.json 2> gpurun_out/g30_$tag.err
  python - <<PY
import json
j=json.load(open("gpurun_out/g30_$tag.json"))
sb=j["setup_breakdown"]
print("$tag: factor %.1f setup %.3f | problem %.3f symbolic %.3f handle %.3f first_step %.3f" % (j["factor_ms"], j["setup_s"], sb["problem_generation_ordering_rhs_s"], sb["symbolic_s"], sb["handle_create_s"], sb["first_step_s"]))
print("   "+" ".join("%s=%.0f" % (k.split(".")[-1][:14], 1e3*v) for k, v in sb["handle_create_phases_s"].items()))
PY
}
run a100 100 A=1
run b100 100 A=1
run c100 100 A=1
run a150 150 A=1
run nothp150 150 SLUAMD_NO_THP=1
