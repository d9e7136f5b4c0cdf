import json, sys
d = json.loads(sys.stdin.read()); m = d["multi_gpu"]
print(sys.argv[1], m["ro_split_pose_equals_unsplit"], m.get("ro_split_check_attempts"), m.get("ro_split_pose_max_abs_diff"), m.get("ro_replica_and_unsplit_pose_spread_over_ranks"),
      m["ray_dp_training"]["params_equal_over_ranks"], m["global_ba_anchor_spread_over_ranks"])
