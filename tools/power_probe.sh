#!/bin/bash
# Socket power and shader clock of GPU 0 while a command runs (read-only rocm-smi polls, 0.5 s apart): tools/power_probe.sh <log> -- <command ...>
# The strip kernels run power-capped: the clock the chip holds (not the pipe cycles a stream needs) sets their MFMA rate -- DESIGN §4 finding 16.
log=$1; shift; shift
"$@" > "$log.cmd.out" 2>&1 &
pid=$!
: > "$log"
while kill -0 $pid 2>/dev/null; do
    rocm-smi -d 0 --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | sed -E 's/\s+/ /g' | tr '\n' '|' >> "$log"
    echo >> "$log"
    sleep 0.5
done
wait $pid
echo "exit $?" >> "$log"
