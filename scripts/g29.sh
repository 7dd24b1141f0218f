mkdir -p gpurun_out
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; ldd --version | head -1; nproc
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-scaling-point --no-configs4 > gpurun_out/g29_$tag.json 2> gpurun_out/g29_$tag.err
  python - <<PY
import json
j=json.load(open("gpurun_out/g29_$tag.json"))
sb=j["setup_breakdown"]
print("$tag: setup %.3f | problem %.3f symbolic %.3f handle %.3f first_step %.3f" % (j["setup_s"], sb["problem_generation_ordering_rhs_s"], sb["symbolic_s"], sb["handle_create_s"], sb["first_step_s"]))
print("   "+" ".join("%s=%.0f" % (k.split(".")[-1][:14], 1e3*v) for k, v in sb["handle_create_phases_s"].items()))
PY
}
run base A=1
run thp GLIBC_TUNABLES=glibc.malloc.hugetlb=1
run base2 A=1
run thp2 GLIBC_TUNABLES=glibc.malloc.hugetlb=1
run t32 SLUAMD_PLAN_THREADS=32
