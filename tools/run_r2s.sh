set -x
O=gpurun_out/r2s; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log); tail -5 $O/pytest.log
timeout 600 python tools/dense_batch_bench.py > $O/dense_batch.json 2> $O/dense_batch.err; cat $O/dense_batch.json
PVS_DENSE_PER_QUERY=1 timeout 600 python tools/dense_batch_bench.py 4000000 > $O/dense_perquery_4M.json 2> $O/dense_perquery.err; cat $O/dense_perquery_4M.json
timeout 600 python tools/dense_batch_bench.py 4000000 > $O/dense_batch_4M.json 2>> $O/dense_batch.err; cat $O/dense_batch_4M.json
timeout 600 python tools/similar_bench.py > $O/similar.json 2> $O/similar.err; cat $O/similar.json
