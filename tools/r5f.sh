#!/bin/bash
set -o pipefail
timeout -k 10 560 bash tools/prof_config_pmc.sh r5 4 2>&1 | tail -12
timeout -k 10 600 bash tools/prof_config_pmc.sh r5 5 2>&1 | tail -12
