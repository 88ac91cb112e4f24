#!/bin/bash
# diagnostics: sample power / clocks with rocm-smi while the recall step runs in a loop
(ITERS=${ITERS:-6000} NQ=${NQ:-256} python tools/step_time.py > /tmp/pp.log 2>&1 &) 
sleep ${WARM:-25}
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.3; done
wait
tail -2 /tmp/pp.log
