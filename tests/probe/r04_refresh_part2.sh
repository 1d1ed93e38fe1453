bash tests/probe/refresh_profiles_r04.sh r4n prof
bash tests/probe/refresh_profiles_r04.sh r4n prof1s
