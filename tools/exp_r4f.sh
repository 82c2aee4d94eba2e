cd $GRAFT_REPO_ROOT
for c in 4 6 8; do
  timeout -k 10 200 python bench.py --workload wd-articles --no-other --no-cpu-baseline --steps 60 --warmup 10 --settle 40 --cfg ark_ce_chunks=$c > gpurun_out/r4f_wda_c$c.json 2> gpurun_out/r4f_wda_c$c.err && python -c "import json; d=json.load(open('gpurun_out/r4f_wda_c$c.json')); print('wd-articles chunks $c ms/step', round(d['ms_per_step'],4))"
done
for ch in 1 2; do
  timeout -k 10 200 python bench.py --no-other --no-cpu-baseline --steps 500 --cfg ark_diag_chains=$ch > gpurun_out/r4f_sp_ch$ch.json 2> gpurun_out/r4f_sp_ch$ch.err && python -c "import json; d=json.load(open('gpurun_out/r4f_sp_ch$ch.json')); print('syn-paths chains $ch ms/step', round(d['ms_per_step'],4))"
done
timeout -k 10 200 python tools/fat_time.py 256 syn-types 2>&1 | tail -4
