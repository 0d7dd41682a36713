cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r02
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02/also_table -o p -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --also-iters 6 > gpurun_out/prof_r02/also_table.log 2>&1 || echo failed
f=$(find gpurun_out/prof_r02/also_table -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/prof_r02/also_table_kernel_stats.csv
head -30 gpurun_out/prof_r02/also_table_kernel_stats.csv | cut -c1-140
