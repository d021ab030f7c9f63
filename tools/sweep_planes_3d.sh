# Developer tool: z-chunk depth of the 3D tile kernels against C4 (513^3, V(2,2) block-Jacobi), one knob at a time.
# usage (GPU box): bash tools/sweep_planes_3d.sh
run() { RUN_CONFIG_REPEATS=3 python tools/run_config.py c4 2>&1 | grep "best of" ; }
echo "default"; run
for p in 8 16 24 32 44 64 88 128 172 256 512; do echo "EVOMG_TWICE_PLANES=$p"; EVOMG_TWICE_PLANES=$p run; done
for p in 4 8 12 16 22 32 43 64 86 128 256; do echo "EVOMG_RR_CPLANES=$p"; EVOMG_RR_CPLANES=$p run; done
for p in 8 16 32 64 128 172 256 512; do echo "EVOMG_STRIP_PLANES=$p (every 3D tile launch)"; EVOMG_STRIP_PLANES=$p run; done
echo "default"; run
