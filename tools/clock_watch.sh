#!/bin/bash
# tools/clock_watch.sh : sample the shader clock / power while bench.py runs (is the hot loop power- or clock-capped?)
cd "$(dirname "$0")/.."
python bench.py --steps 3000 --warmup 20 > /tmp/cw_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*sclk clock level: //' -e 's/.*Power (W): /P=/' | tr '\n' ' '
  echo
  sleep 0.5
done | sort | uniq -c | sort -k1,1nr | head -30
tail -1 /tmp/cw_bench.log | cut -c1-200
