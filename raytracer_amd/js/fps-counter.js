'use strict';
/*
 * Frames-per-second meter with the interface of the reference's FPSCounter
 * (src/fps-counter.ts): updateFPS() once per loop tick, getFPS() for the last
 * completed window.  A window closes at the first tick that comes at least one
 * second after it opened; its rate is ticks * 1000 / elapsed ms, rounded.
 */
const clock = require('perf_hooks').performance;

const WINDOW_MS = 1000;

function FPSCounter() {
    let windowStart = clock.now();
    let ticks = 0;
    let rate = 0;

    this.updateFPS = function () {
        ticks += 1;
        const elapsed = clock.now() - windowStart;
        if (elapsed < WINDOW_MS) return;
        rate = Math.round(ticks * WINDOW_MS / elapsed);
        windowStart += elapsed;
        ticks = 0;
    };

    this.getFPS = function () {
        return rate;
    };
}

module.exports = { FPSCounter };
