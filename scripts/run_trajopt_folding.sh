python -m thinshelllab_amd.training.trajopt_folding --l 12 --r 13 --iter 400 --tot_step 50 --lr 0.00003 --curve7 1 --curve8 -1
