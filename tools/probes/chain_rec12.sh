# GPU box: the range coder's loop with 12-byte records (tools/gen_chain_asm.py) through tools/ubench_chain_f64.hip: exactness of
# every variant against the integer recurrence + clocks per symbol, by symbols a lane takes in a row and with parts left out
cd /root/repo
for v in "product:" "per8:GZ_GEN_PER=8" "per16:GZ_GEN_PER=16" "per24:GZ_GEN_PER=24" "no_checkpoints:GZ_GEN_CKPT=0" "no_hop:GZ_GEN_HOP=0" "no_prep:GZ_GEN_PREP=0" "bare:GZ_GEN_HOP=0 GZ_GEN_CKPT=0 GZ_GEN_PREP=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python tools/gen_chain_asm.py /tmp/chain_$name.h
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I tools -DGZ_CHAIN_HDR="\"/tmp/chain_$name.h\"" tools/ubench_chain_f64.hip -o /tmp/ub_$name 2>/dev/null
    echo "== $name"
    timeout 120 /tmp/ub_$name 2>&1 | grep "hop: \|exactness \[hop" | head -5
done
