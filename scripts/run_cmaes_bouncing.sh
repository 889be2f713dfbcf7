python -m thinshelllab_amd.training.run_cmaes_parameter --tot_step 30 --iter 5 --trial 1 --pop_size 10 --sigma 0.2 --env bouncing --Kb 100 --mu 0.5
