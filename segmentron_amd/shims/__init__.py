"""Stand-ins for third-party packages the reference imports but the ROCm image lacks
(SURVEY.md F1).  This directory is appended to sys.path by segmentron_amd.dropin.install_shims()
only when the real package is absent."""
