python -m thinshelllab_amd.training.run_cmaes_all --abs_step 5 --tot_step 50 --iter 15 --trial 1 --pop_size 40 --sigma 1.0 --env lifting --Kb 100 --mu 5.0 --max_dist 0.001
