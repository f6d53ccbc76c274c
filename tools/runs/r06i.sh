cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -n 3 gpurun_out/r06_profile_round.log | cut -c1-300
bash tools/pmc_traffic.sh r06_m6w256 256 f16m6 > gpurun_out/r06_pmc.log 2>&1
tail -n 4 gpurun_out/r06_pmc.log
