#!/bin/bash
# counterpart of /root/reference/code/scripts/run_trajopt_separating.sh (empty in the reference)
python -m thinshelllab_amd.training.trajopt_interact --l 0 --r 1 --iter 400 --tot_step 50 --lr 0.00003 --sep
