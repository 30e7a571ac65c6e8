#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the default bench command (attention section), round 3 state "a"
tools/prof_pmc.sh r03a > gpurun_out/r3/prof_r03a.log 2>&1
tail -80 gpurun_out/prof_r03a/summary.md
