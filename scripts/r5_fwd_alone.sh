# the streaming forward alone (scripts/fwd_alone.py) for several builds of the library: LIBS="name=path ..." (empty path = in-tree)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_fwd_alone
mkdir -p $O
for kv in $LIBS; do
  name=${kv%%=*}; lib=${kv#*=}
  for items in 25024 200000; do
    for v in $VARIANTS; do
      LTG_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/scripts/fwd_alone.py $items 200 $v > $O/log_${name}_${items}_$v.txt 2>&1
      f=$(find $O/p -name "*kernel_stats.csv" | head -1)
      echo "== $name items $items variant $v: $(grep 'us per forward' $O/log_${name}_${items}_$v.txt)"
      grep -i "dec1_fwd_stream\|row_stats\|fk_dec0\|fk_enc" "$f" | awk -F'","' '{print $1" calls "$2" avg_ns "$4" min "$6" max "$7}' | cut -c1-160
      rm -rf $O/p
    done
  done
done
