set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/e; mkdir -p $O
T=tests/test_gpu_fresh_orderings.py::test_signatures_are_captured_on_their_second_sighting
python -X faulthandler -m pytest $T -x -q > $O/a_default.log 2>&1; echo "default rc=$?"
BBD_EXPERIMENT=1 BBD_IDENT_GROUPED=1 python -X faulthandler -m pytest $T -x -q > $O/b_grouped_ident.log 2>&1; echo "grouped ident rc=$?"
BBD_GRAPH_SHARED_POOL=0 python -X faulthandler -m pytest $T -x -q > $O/c_private_pool.log 2>&1; echo "private pool rc=$?"
BBD_MIOPEN_DB=0 python -X faulthandler -m pytest $T -x -q > $O/d_no_db.log 2>&1; echo "no db rc=$?"
BBD_GRAPH_SHARED_POOL=0 BBD_EXPERIMENT=1 BBD_IDENT_GROUPED=1 python -X faulthandler -m pytest $T -x -q > $O/e_both.log 2>&1; echo "private+grouped rc=$?"
tail -3 $O/*.log
