# the round's profile set on the FINAL build: bash scripts/profile_round4.sh r04   (GPU box, repo root) -> gpurun_out/<tag>_*; copy into profiles/
R=$GRAFT_REPO_ROOT
tag=${1:-r04}
bash $R/scripts/profile_round.sh $tag > $R/gpurun_out/${tag}_profile_round.log 2>&1            # kernel trace + stats + SQ + FETCH/WRITE passes
PASS_TIMEOUT=200 bash $R/scripts/pmc_requests.sh ${tag}_req TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_EA0_RDREQ_LEVEL_sum > $R/gpurun_out/${tag}_req.log 2>&1
python3 $R/scripts/pmc_requests_json.py $R/gpurun_out/${tag}_requests.json $R/gpurun_out/${tag}_req/summary.txt
bash $R/scripts/profile_cfg.sh $tag cfg4 > /dev/null 2>&1
bash $R/scripts/profile_cfg.sh $tag cfg5 > /dev/null 2>&1
bash $R/scripts/pmc_sq.sh ${tag}_cfg4_pmc_sq cfg4 > $R/gpurun_out/${tag}_cfg4_pmc_sq.txt 2>&1
bash $R/scripts/pmc_sq.sh ${tag}_cfg5_pmc_sq cfg5 > $R/gpurun_out/${tag}_cfg5_pmc_sq.txt 2>&1
bash $R/scripts/fastq_trace.sh $tag 8000000 16 > $R/gpurun_out/${tag}_fastq_trace_run.log 2>&1
ls -la $R/gpurun_out/${tag}_*
