"""Unit constants in the reference's (R_sun, M_sun, day) system.

Values are the literal fall-backs the reference ships for when astropy is not
usable (/root/reference/src/exoplanet/orbits/constants.py:32-37); astropy is
not a dependency here."""

G_grav = 2942.2062175044193          # R_sun^3 / M_sun / day^2
gcc_per_sun = 5.905271918964842      # (M_sun / R_sun^3) in g / cm^3
au_per_R_sun = 0.00465046726096215
c_light = 37231.66360672704          # R_sun / day
day_per_yr_over_2pi = 58.13244087623438
# (1 R_sun / day) in m / s, for get_radial_velocity's default output unit
m_per_s_per_Rsun_per_day = 695700000.0 / 86400.0
