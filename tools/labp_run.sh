for b in tools/labp_ilv.bin tools/labp_m16.bin; do
for s in "20480 1152 384" "20480 384 384" "20480 1536 384" "20480 384 1536" "20480 384 1152" "10240 1152 384" "10240 1536 384" "10240 384 1536"; do
 timeout -k 5 30 $b $s 50 || exit 1
done; done
