bash scripts/gpu_sort.sh
W=triplet bash scripts/gpu_prof_triplet.sh | tail -6
