#!/bin/bash
# Samples power / clocks / temperature with rocm-smi while a command runs:  tools/smi_sample.sh <out.log> <cmd...>
out=$1; shift
"$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|Temperature (Sensor junction)\|socclk" | tr '\n' ';' >> "$out"
  echo >> "$out"
  sleep 0.5
done
wait $pid
