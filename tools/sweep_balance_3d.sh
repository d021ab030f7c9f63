# Developer tool: z-chunk depth of the 3D tiles chosen per level by workgroup balance (EVOMG_BALANCE3D = deepest chunk allowed).
run() { RUN_CONFIG_REPEATS=3 python tools/run_config.py c4 2>&1 | grep "best of" ; }
echo "default"; run
for cap in 1 32 48 64 96 128 176; do echo "EVOMG_BALANCE3D=$cap"; EVOMG_BALANCE3D=$cap run; done
echo "default"; run
for cap in 48 64; do echo "EVOMG_BALANCE3D=$cap"; EVOMG_BALANCE3D=$cap run; done
