#!/bin/bash
# every differential fuzzer on fresh seed ranges, one lease:  gpurun --timeout 2400 -- 'bash profiles/fuzz_campaign.sh'
# BASE / BASE_LARGE move the seed ranges, S_* the number of seeds per fuzzer.
# (seed ranges disjoint from the tests' and from the earlier rounds' runs: FIRST below)  -> gpurun_out/fuzz_campaign.txt
OUT=gpurun_out/fuzz_campaign.txt
mkdir -p gpurun_out
: > $OUT
run() { echo "== $*" | tee -a $OUT; ( time env "$@" ) 2>&1 | grep -v "RuntimeWarning\|return (d \* r" | tail -${TAIL:-6} | cut -c1-1500 | tee -a $OUT; }
run SEEDS=${S_REG:-300} FIRST=${BASE:-50000} timeout 900 python profiles/fuzz_reg.py
run SEEDS=${S_SAMPLING:-200} FIRST=${BASE:-50000} timeout 900 python profiles/fuzz_sampling.py
run SEEDS=${S_MULTI:-150} FIRST=${BASE:-50000} timeout 900 python profiles/fuzz_multi.py
run SEEDS=${S_OVERLAP:-300} FIRST=${BASE:-50000} timeout 600 python profiles/fuzz_overlap.py
run SEEDS=${S_TSDF:-80} FIRST=${BASE:-50000} timeout 900 python profiles/fuzz_tsdf.py
run SEEDS=${S_TSDF_BIG:-12} FIRST=$((${BASE:-50000} + 10000)) BIG=1 timeout 900 python profiles/fuzz_tsdf.py
run SEEDS=${S_REPLAY:-200} FIRST=${BASE:-50000} timeout 900 python profiles/fuzz_replay.py
run SEEDS=${S_REPLAY_BIG:-40} FIRST=$((${BASE:-50000} + 10000)) BIG=1 timeout 900 python profiles/fuzz_replay.py
run SEEDS=${S_REG_LARGE:-10} FIRST=${BASE_LARGE:-500} timeout 900 python profiles/fuzz_reg_large.py
run SEEDS=${S_SAMPLING_LARGE:-5} FIRST=${BASE_LARGE:-500} timeout 900 python profiles/fuzz_sampling_large.py
