#!/bin/bash
# counterpart of /root/reference/code/scripts/run_dp_slide.sh (empty in the reference)
python -m thinshelllab_amd.training.trajopt_sliding --l 0 --r 1 --iter 50 --tot_step 50 --lr 0.001 --mu 1.0
