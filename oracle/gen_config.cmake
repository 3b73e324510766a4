# Instantiates the reference's OWN config template (src/config.hh.in) with cmake's
# configure_file(), exactly the step the reference's top-level CMakeLists.txt performs
# (CMakeLists.txt:12-14 version numbers, :27-44 feature switches), without running that
# CMakeLists (it does not configure in this image: SURVEY.md §8c).
#
# Feature switches are left UNSET on purpose: FFTW3, PortAudio and librtlsdr are not
# installed here, so find_package() would leave them unset too.
#
# usage: cmake -DREF=/root/reference -DOUT=oracle/_ref -P oracle/gen_config.cmake
set(libsdr_VERSION_MAJOR "0")
set(libsdr_VERSION_MINOR "1")
set(libsdr_VERSION_PATCH "0")
configure_file(${REF}/src/config.hh.in ${OUT}/config.hh)
