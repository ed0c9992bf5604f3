// placeholder, filled in later (simple-knn / operate_points / stereo_vision oracle)
