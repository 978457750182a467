ls /sys/class/drm/ 2>/dev/null | head; ls /sys/class/drm/card*/device/ 2>/dev/null | grep -i "gpu_metrics\|pp_dpm_sclk\|hwmon" | head
for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f; done 2>/dev/null | head -20
ls /sys/class/drm/card*/device/hwmon/*/ 2>/dev/null | head -30
(python tools/two_streams.py -1 > /dev/null 2>&1 &) ; sleep 6
for i in 1 2 3; do cat /sys/class/drm/card*/device/hwmon/*/freq1_input 2>/dev/null | head -3; rocm-smi --showclocks 2>&1 | grep -i "sclk\|fclk" | head -4; sleep 0.5; done
which amd-smi && timeout 20 amd-smi metric --clock 2>&1 | head -40
wait
