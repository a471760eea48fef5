# build esrecsys_amd/libesr_hip_<name>.so with extra compile flags: bash scripts/build_variant.sh <name> <flags...>
name=$1; shift
mkdir -p /tmp/var_$name
cd esrecsys_amd/csrc
for f in *.hip; do
  [ "$f" = "esr_probe.hip" ] && continue
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I../../include -c $f -o /tmp/var_$name/${f%.hip}.o 2>/tmp/var_$name/${f%.hip}.log || echo "FAILED $f" ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libesr_hip_$name.so /tmp/var_$name/*.o -ldl && echo "built libesr_hip_$name.so"
