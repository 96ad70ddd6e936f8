#!/bin/bash
# Round 4, after the walk tables (K3t / K4t): the drop-in CLIs against the compiled reference with the tables forced on at any size
# (TSTA_WALK_TABLE=2) and at their defaults on long inputs.  usage (GPU box): bash tools/scripts/r04_soak_walk_tables.sh <tag>
O=gpurun_out/$1
mkdir -p $O
{
echo "== TSTA_psa vs reference, 40 random pairs (200-20 000 bp, four kinds, five scorings), TSTA_WALK_TABLE=2"
TSTA_WALK_TABLE=2 timeout -k 10 600 python tests/tools/soak_psa_vs_reference.py 4242 40 2>&1 | tail -3
echo "== TSTA_msa vs reference, 12 long families (3-5 sequences of 4-20 kbp), defaults (tables on from 1 536 columns)"
timeout -k 10 900 python tests/tools/soak_msa_long.py 77 12 2>&1 | tail -3
echo "== TSTA_msa vs reference, 300 structurally varied families (80-2 600 bp), default scoring, -W 10, TSTA_WALK_TABLE=2"
TSTA_WALK_TABLE=2 timeout -k 10 900 python tests/tools/soak_msa_campaign.py "2,-5,-2,-4" 10 300 52000 2>&1 | tail -3
echo "== the same with tables off (control)"
TSTA_WALK_TABLE=0 timeout -k 10 600 python tests/tools/soak_msa_campaign.py "2,-5,-2,-4" 10 100 52000 2>&1 | tail -2
} > $O/soak_walk_tables.txt 2>&1
tail -20 $O/soak_walk_tables.txt
