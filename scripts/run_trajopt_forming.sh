#!/bin/bash
# counterpart of /root/reference/code/scripts/run_trajopt_forming.sh (empty in the reference)
python -m thinshelllab_amd.training.trajopt_forming --l 0 --r 1 --iter 400 --tot_step 50 --lr 0.00003 --target_dir ${1:?path to cloth_pos.npy}
