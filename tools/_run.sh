python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "edge or fuzz" 2>&1 | tail -3
