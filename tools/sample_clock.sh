#!/bin/bash
# sample the shader clock and the socket power while the headline loop runs
(timeout 120 python bench.py --steps 1500 --warmup 3 --no-cpu --no-secondary > /tmp/bench_clk.json 2>/dev/null) &
sleep 14
for i in $(seq 1 12); do
  for c in /sys/class/drm/card*/device; do
    [ -f $c/pp_dpm_sclk ] && echo "sclk: $(grep '\*' $c/pp_dpm_sclk | tr '\n' ' ')" 
    for h in $c/hwmon/hwmon*; do
      [ -f $h/power1_average ] && echo "power_uW: $(cat $h/power1_average)"
      [ -f $h/power1_input ] && echo "power_in_uW: $(cat $h/power1_input)"
      [ -f $h/freq1_input ] && echo "freq1_Hz: $(cat $h/freq1_input)"
      [ -f $h/power1_cap ] && echo "cap_uW: $(cat $h/power1_cap)"
    done
  done
  sleep 0.3
done
wait
tail -c 600 /tmp/bench_clk.json
echo
rocm-smi --showclocks --showpower 2>/dev/null | head -30
