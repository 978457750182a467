O=gpurun_out/foot16_ab.txt; : > $O
for rep in 1 2 3; do
python tools/ab_frame.py 2>&1 | grep -v amdgpu.ids >> $O
CSKY_LIBRARY=$PWD/godot-volumetric-cloud-demo-v2_amd/libcloudsky_foot16.so python tools/ab_frame.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
