# rocprofv3 --kernel-trace --stats per BASELINE configuration (ROUND=r04 by default): the kernel-stats CSVs that back bench.py's `also` table.
# Copies <name>_kernel_stats.csv into gpurun_out/prof_${ROUND:-r05}/; commit them under profiles/ as <round>_<name>_kernel_stats.csv.
: ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_${ROUND:-r05}
run() { # name, command...
  name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${ROUND:-r05}/$name -o p -- "$@" > gpurun_out/prof_${ROUND:-r05}/$name.log 2>&1 || echo "$name failed"
  f=$(find gpurun_out/prof_${ROUND:-r05}/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/prof_${ROUND:-r05}/${name}_kernel_stats.csv
}
run bench_planar python3 bench.py --no-cpu-baseline --no-also --steps 50 --warmup 5
run bench_rgba python3 bench.py --no-cpu-baseline --no-also --steps 50 --warmup 5 --layout rgba
run config3 python3 tools/run_p3.py 3 planar 20
run config4_rank python3 tools/run_p3.py 4 planar 20
run config5_fixed python3 tools/run_p3.py 5 planar 10
run config5_focus_map python3 tools/run_focus.py auto 15 3840 2160 scene
run config2_focus_map python3 tools/run_focus.py auto 8 1920 1080 scene
run also_table python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --also-iters 6
run config5_allfocus_ten python3 tools/run_allfocus.py TEN_WM 6 estimated
run config5_allfocus_std python3 tools/run_allfocus.py STD 6 estimated
run config5_allfocus_std_once python3 tools/run_allfocus.py STD 6 estimated filtered_gather_once
python3 bench.py --no-cpu-baseline --no-also --steps 50 --warmup 5 > /dev/null 2>&1; head -12 gpurun_out/prof_${ROUND:-r05}/bench_planar_kernel_stats.csv
