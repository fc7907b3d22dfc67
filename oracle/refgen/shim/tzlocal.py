def get_localzone():
    import datetime
    return datetime.timezone.utc
