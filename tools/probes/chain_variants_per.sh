# GPU box: what each part of the range coder's loop costs - the generated loop with parts left out (timing only: results of the
# variants are wrong by construction), through tools/ubench_chain_f64.hip
set -e
cd /root/repo
for v in "product:" "touch:GZ_GEN_TOUCH=1" "bare_touch:GZ_GEN_TOUCH=1 GZ_GEN_HOP=0 GZ_GEN_CKPT=0 GZ_GEN_PREP=0" "bare:GZ_GEN_HOP=0 GZ_GEN_CKPT=0 GZ_GEN_PREP=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python tools/gen_chain_asm.py /tmp/chain_$name.h
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I tools -DGZ_CHAIN_HDR="\"/tmp/chain_$name.h\"" tools/ubench_chain_f64.hip -o /tmp/ub_$name 2>/dev/null
    echo "== $name"
    timeout 120 /tmp/ub_$name 2>&1 | grep "hop:  1 chains\|exactness \[hop" | grep "hop:  1 chains\|exactness \[hop, data A" | head -2
done
