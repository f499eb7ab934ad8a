#!/bin/bash
# Y355_RING_HALF = (stagger in 64-cycle units) << 8 | mask (1: conv6/7, 2: conv5)
for v in 0 1 3 $((48*256+1)) $((80*256+1)) $((112*256+1)) $((160*256+1)) $((80*256+3)) $((112*256+3)); do
  echo "== HALF=$v (mask $((v & 255)) stagger $((v >> 8)) x 64 cycles)"
  Y355_RING_HALF=$v bash scratch/quick.sh
done
