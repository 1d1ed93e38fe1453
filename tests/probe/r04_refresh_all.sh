# the whole round-4 measurement pass in one call: PMC + rocprof passes first (the bench line reads their tables from profiles/ only on the
# NEXT call, so this script is followed by collect_r04.sh and tests/probe/r04_final.sh)
bash tests/probe/refresh_profiles_r04.sh r4n prof
bash tests/probe/refresh_profiles_r04.sh r4n prof1s
bash tests/probe/refresh_profiles_r04.sh r4n clock
