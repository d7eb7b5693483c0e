python -m pytest tests/test_gpu_restorers.py -m gpu -x -q 2>&1 | tail -5
python bench.py --slot dct --frames 6 2>/dev/null
ELVIS_DCN_GENERIC=1 python bench.py --slot dct --frames 6 2>/dev/null
python bench.py --slot dct --frames 6 2>/dev/null
