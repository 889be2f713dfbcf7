#!/bin/bash
# counterpart of /root/reference/code/scripts/run_dp_bouncing.sh (empty in the reference; flags of run_dp_card.sh's family)
python -m thinshelllab_amd.training.trajopt_bouncing --l 0 --r 1 --iter 50 --tot_step 80 --lr 20000 --Kb 1400
