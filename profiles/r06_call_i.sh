#!/bin/bash
# round 6, GPU call I: the driver's three commands on the final tree
bash tools/driver_commands.sh r06_zzz
