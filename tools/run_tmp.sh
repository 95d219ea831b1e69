cd /root/repo
python tools/tools_time.py c3 30 | grep -E "TIME|nodes" | cut -c1-700
python tools/tools_time.py c4 10 | grep -E "TIME|nodes" | cut -c1-900
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or sampled_against_c_oracle" 2>&1 | tail -3
