for b in tools/labp_ilv.bin tools/labp_var3.bin; do
for s in "20480 1152 384" "20480 384 1536" "10240 1152 384" "10240 384 1536"; do
 timeout -k 5 30 $b $s 50 || exit 1
done; done
