#!/bin/bash
# round 6, GPU call P: the driver's three commands on the LAST tree
bash tools/driver_commands.sh r06_final
