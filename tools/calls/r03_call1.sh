O=gpurun_out/r03a; mkdir -p $O
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > $O/pytest.log 2>&1
timeout 120 python tools/sampler_timeline.py > $O/sampler_timeline.txt 2>&1
timeout 300 python tools/boundary_probe.py > $O/boundary_probe.txt 2>&1
timeout 600 python tools/ar_tile_sweep.py > $O/ar_tile_sweep.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/ar_probe.py 32 200 > $GRAFT_REPO_ROOT/$O/ar_probe_traced.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $f > $O/trace_gaps_whole_chip.txt 2>&1
timeout 120 python tools/ar_probe.py 32 200 > $O/ar_probe.txt 2>&1
tail -5 $O/pytest.log; cat $O/sampler_timeline.txt; cat $O/boundary_probe.txt; cat $O/ar_tile_sweep.txt; cat $O/trace_gaps_whole_chip.txt; cat $O/ar_probe.txt
