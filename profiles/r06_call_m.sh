#!/bin/bash
# round 6, GPU call M: the driver's three commands on the last tree
bash tools/driver_commands.sh r06_zzzzz
