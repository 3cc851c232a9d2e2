for spin in 0 10 25; do timeout 60 tools/handoff_probe $spin 4; done
timeout 60 tools/handoff_probe 10 8
timeout 60 tools/handoff_probe 10 1
