// Stand-in for plotly.js handed to kaleido's headless Chromium: "renders" a figure whose first
// trace carries an Ogg/Theora file (base64) by decoding it with the browser's own decoder and
// returning the RGBA pixels of the requested frames.
window.Plotly = {
  version: '9.9.9',
  toImage: function(fig, opts) {
    return new Promise(function(resolve, reject) {
      var req = fig.data[0];
      var v = document.createElement('video');
      v.muted = true; v.preload = 'auto';
      var out = {frames: [], log: []};
      var fail = function(m) { out.error = m; resolve(JSON.stringify(out)); };
      v.onerror = function() { fail('video error ' + (v.error ? v.error.code + ' ' + v.error.message : '?')); };
      v.onloadeddata = function() {
        out.w = v.videoWidth; out.h = v.videoHeight; out.duration = v.duration;
        var c = document.createElement('canvas'); c.width = v.videoWidth; c.height = v.videoHeight;
        var ctx = c.getContext('2d');
        var i = 0;
        var step = function() {
          if (i >= req.nframes) { resolve(JSON.stringify(out)); return; }
          v.onseeked = function() {
            ctx.drawImage(v, 0, 0);
            var d = ctx.getImageData(0, 0, c.width, c.height).data;
            var s = '';
            for (var k = 0; k < d.length; k += 4) s += String.fromCharCode(d[k], d[k + 1], d[k + 2]);
            out.frames.push(btoa(s));
            out.log.push(v.currentTime);
            i++;
            step();
          };
          v.currentTime = (i + 0.5) / req.fps;
        };
        step();
      };
      v.src = 'data:video/ogg;base64,' + req.ogv;
      setTimeout(function() { fail('timeout; readyState ' + v.readyState); }, 20000);
    });
  }
};
