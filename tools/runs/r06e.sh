cd $GRAFT_REPO_ROOT
bash tools/trace_step.sh 1024 1024 r06e_1024slots --dtype f16m6 > gpurun_out/r06e_trace.log 2>&1
bash tools/trace_step.sh 8 8 r06e_large8 --dtype f16m6 >> gpurun_out/r06e_trace.log 2>&1
bash tools/prof_quick.sh r06e_dec1024 --dtype f16m6 --windows 1024 --decode-only --iters 2 > gpurun_out/r06e_prof.log 2>&1
tail -n 32 gpurun_out/r06e_prof.log
