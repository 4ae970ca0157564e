VARIANT_SRC=exo_celerite.hip tools/build_variant.sh gp_noflag -DEXO_GP_COND_MAX=1e30
