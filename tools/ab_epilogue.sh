set -e
python -m pytest tests/test_gpu_trk.py -x -q -k "epilogue_forms or other_block_lengths or replay_reproduces or batched_receivers" > gpurun_out/e8_test.log 2>&1 || { tail -30 gpurun_out/e8_test.log; exit 1; }
tail -2 gpurun_out/e8_test.log
for i in 1 2; do
  GPSMI_EPILOGUE_FORM=1 python bench.py --no-extra --no-cpu > gpurun_out/e8_new$i.json 2>gpurun_out/e8_new$i.err
  GPSMI_EPILOGUE_FORM=0 python bench.py --no-extra --no-cpu > gpurun_out/e8_old$i.json 2>gpurun_out/e8_old$i.err
done
python tools/pick_line.py gpurun_out/e8_new1.json gpurun_out/e8_old1.json gpurun_out/e8_new2.json gpurun_out/e8_old2.json
