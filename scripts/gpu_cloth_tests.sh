cd $GRAFT_REPO_ROOT
make -C oracle >/dev/null 2>&1
python -m pytest tests/test_gpu_cloth.py -m gpu -x -q 2>&1 | tail -30
