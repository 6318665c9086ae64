P=$PWD/state_policy_diffusionmodel_amd/libspdm_prev.so
for sw in "X=1" "SPDM_NO_SA_FUSED=1" "SPDM_NO_SA_TAIL=1" "SPDM_NO_SKINNY=1" "SPDM_NO_SA_FUSED=1 SPDM_NO_SA_TAIL=1" "SPDM_NO_SA_FUSED=1 SPDM_NO_SA_TAIL=1 SPDM_NO_SKINNY=1 SPDM_NO_WIDE=1" "SPDM_NO_SA_FUSED=1 SPDM_NO_SA_TAIL=1 SPDM_ATTN_VALU=1"; do
  a=$(env $sw python tools/probes/same_bits.py 1 2>/dev/null | tail -1)
  b=$(env $sw SPDM_LIB=$P python tools/probes/same_bits.py 1 2>/dev/null | tail -1)
  echo "$sw | new $a | prev $b"
done
