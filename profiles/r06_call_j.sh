#!/bin/bash
# round 6, GPU call J: the driver's three commands once more on the last tree
# (after the self-review fixes of the producer's re-binding state)
bash tools/driver_commands.sh r06_zzzz
