# A/B of factorisation-path switches on one rank's share (4 latents) and the whole C2 job (32); runs ON THE GPU BOX
OUT=gpurun_out/${AB_TAG:-ab}
mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python tools/share_profile.py 8 1 > $OUT/share_$name.log 2>&1 || return 1
  env "$@" LMM_PROF_DUMP=1 timeout -k 10 200 python tools/share_profile.py 8 2> $OUT/dump_$name.txt >/dev/null || return 1
  python tools/prof_by_level.py $OUT/dump_$name.txt > $OUT/levels_$name.txt
  grep "cls=1 " $OUT/dump_$name.txt > $OUT/launches_$name.txt
  rm -f $OUT/dump_$name.txt
  echo "== $name"; grep world $OUT/share_$name.log
}
run tail0 LMM_TAIL_POLICY=0 && run tail1 LMM_TAIL_POLICY=1 && run tail1_nofuse LMM_TAIL_POLICY=1 LMM_FUSE_BULK=0
