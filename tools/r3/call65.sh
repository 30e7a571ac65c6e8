#!/bin/bash
tools/prof_layer_pmc.sh r3 2>&1 | tail -40
