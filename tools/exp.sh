python -m pytest tests -m gpu -q -x -k "match" 2>&1 | tail -3
python tools/match_bench.py 2>&1 | tail -2
