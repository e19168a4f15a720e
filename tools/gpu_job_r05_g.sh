set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/g; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputests.log; tail -3 $O/gputests.log
REPS="1 2" bash tools/instep_ab.sh "md2 boosted boosted15_coherent" nopresent present > $O/present_ab.txt 2>&1; cat $O/present_ab.txt
timeout 600 python -X faulthandler bench.py --config md2_loader --no-cpu-baseline --no-secondary > $O/bench_md2_loader.json 2> $O/bench_md2_loader.err; echo "loader rc=$?"; cat $O/bench_md2_loader.json | cut -c1-1500
bash tools/layout_ab.sh > $O/layout_ab.txt 2>&1; cat $O/layout_ab.txt
