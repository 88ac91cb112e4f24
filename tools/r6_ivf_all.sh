bash tools/r6_ivf_ab.sh; bash tools/r6_ivf_phases.sh
