python -m thinshelllab_amd.training.trajopt_lifting --l 0 --r 1 --iter 400 --tot_step 50 --lr 0.00001
