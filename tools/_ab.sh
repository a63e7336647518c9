L=multimodal-aspect-category-sentiment-analysis_amd/fcmf_framework/libfcmf_hip.so
cp $L /tmp/prod.so
run() { python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; }
echo product; run
for v in ln_z ln_zy ln_zdz ln_all; do cp tools/bin/$v/libfcmf_hip.so $L; echo $v; run; run; done
cp /tmp/prod.so $L; echo product; run
