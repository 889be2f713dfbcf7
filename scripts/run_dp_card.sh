#!/bin/bash
# counterpart of /root/reference/code/scripts/run_dp_card.sh
python -m thinshelllab_amd.training.trajopt_card --l 0 --r 1 --iter 50 --tot_step 80 --lr 20000 --Kb 1400
