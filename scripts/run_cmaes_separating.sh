python -m thinshelllab_amd.training.run_cmaes_all --abs_step 5 --tot_step 30 --iter 10 --trial separate --pop_size 40 --sigma 2.0 --env interact --Kb 100 --mu 5.0 --dense 20000.0
