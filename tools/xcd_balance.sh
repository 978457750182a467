make -C godot-volumetric-cloud-demo-v2_amd/csrc timeline -s > /dev/null 2>&1
export CSKY_LIBRARY=$PWD/godot-volumetric-cloud-demo-v2_amd/libcloudsky_timeline.so
for sch in -1 5 2 7; do for rep in 1 2; do echo "== 1/8 share schedule $sch rep $rep"; python tools/timeline.py 8 $sch 2>/dev/null | grep -E "^launch|XCD|integral"; done; done
echo "== 1/4 share auto"; python tools/timeline.py 4 -1 2>/dev/null | grep -E "^launch|XCD|integral"
rm -f godot-volumetric-cloud-demo-v2_amd/libcloudsky_timeline.so
