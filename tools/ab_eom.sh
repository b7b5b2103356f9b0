# same-box A/B of the stacked sigma build: the tree's library against another build (PYMES_AMD_LIBRARY), alternating.
# The other build, from a commit:   rm -rf _ab && mkdir _ab && git archive <commit> pymes_amd/csrc include | tar -x -C _ab/ &&
#                                   make -C _ab/pymes_amd/csrc -j4        (_ab/ is git-ignored; it travels to the GPU box)
for i in 1 2 3; do
  REPS=20 PYMES_AMD_LIBRARY=$PWD/_ab/pymes_amd/lib/libpymes_amd.so timeout -k 10 200 python tools/eom_prof_many.py | head -1 | sed 's/^/old: /'
  REPS=20 timeout -k 10 200 python tools/eom_prof_many.py | head -1 | sed 's/^/new: /'
done
